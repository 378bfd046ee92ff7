#!/bin/bash
# tools/r04_final.sh <part> <git-sha> — the evidence set of round 4 (run on the GPU box from the repo root); outputs under gpurun_out/r04_final/
#   part A: GPU test suite on the shipped and on the lab library, the default bench line (with configs_alt), C host with --verify
#   part B: rocprofv3 --kernel-trace --stats of the bench command (default fusion and one launch per layer), PMC traffic passes -> traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
PART=${1:-A}
SHA=${2:-unknown}
mkdir -p $O
cd $R
M=$R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet
if [ "$PART" = "A" ]; then
  python -m pytest tests -m gpu -q > $O/pytest_gpu_lean.log 2>&1; tail -2 $O/pytest_gpu_lean.log
  MBN_LAB=1 python -m pytest tests -m gpu -q > $O/pytest_gpu_lab.log 2>&1; tail -2 $O/pytest_gpu_lab.log
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  $M --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 --streams 2 --verify > $O/c_host_gpus1.txt 2>&1; tail -4 $O/c_host_gpus1.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
else
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unf -- python3 $R/bench.py --streams 1 --no-fuse-stem --fuse-blocks 0 --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt > $O/bench_unfused_under_rocprof.json 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_unf/*/*kernel_stats.csv | head -1) $O/kernel_stats_unfused.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -- python3 $R/bench.py --dtype bf16 --batch 512 --no-cpu-baseline --no-unfused-stages > $O/bench_bf16_under_rocprof.json 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_bf16/*/*kernel_stats.csv | head -1) $O/kernel_stats_bf16.csv
  rm -rf $O/stats $O/stats_unf $O/stats_bf16
  echo "stats done"
  cd $R
  PW=3,5,7,9,11,13,15,17,19,21,23,25,27
  DW=2,4,6,8,10,12,14,16,18,20,22,24,26
  pass() {   # tag counter layer-list extra-args...
    local tag=$1 ctr=$2 layers=$3; shift 3
    bash $R/tools/pmc_pass.sh r04t_${tag} $ctr -- --layers $layers --iters 3 --warmup 1 "$@" > $O/pmc_${tag}.log 2>&1
    cp $(ls $R/gpurun_out/pmc_r04t_${tag}/*/*counter_collection.csv | head -1) $O/${tag}_counter_collection.csv
    echo "pass $tag done"
  }
  for cfg in "f32|--batch 256" "bf16_1x224|--batch 512 --dtype bf16" "bf16_0.5x160|--batch 512 --dtype bf16 --alpha 0.5 --res 160"; do
    key=${cfg%%|*}; args=${cfg#*|}
    pass ${key}_pwF FETCH_SIZE $PW $args
    pass ${key}_pwW WRITE_SIZE $PW $args
    pass ${key}_dwF FETCH_SIZE $DW $args
    pass ${key}_dwW WRITE_SIZE $DW $args
    python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r04t_${key}_pwF $R/gpurun_out/pmc_r04t_${key}_pwW $PW 4 > $O/${key}_pw.json
    python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r04t_${key}_dwF $R/gpurun_out/pmc_r04t_${key}_dwW $DW 4 > $O/${key}_dw.json
  done
  python3 $R/tools/make_traffic.py assemble $O $SHA > $O/traffic.json
  echo "traffic.json written"
fi
echo done

#!/bin/bash
# tools/r05_final.sh <part> — the evidence set of round 5 (run on the GPU box from the repo root); outputs under gpurun_out/r05_final/
#   part A: the default bench line exactly as the driver runs it (+ its side file), GPU test suite on the shipped and on the lab library, C host with --verify, smoke
#   part B: rocprofv3 --kernel-trace --stats of the bench command (default fusion, one launch per layer, bf16)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
PART=${1:-A}
mkdir -p $O
cd $R
M=$R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet
if [ "$PART" = "A" ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --record $O/bench_full_record.json > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$? bytes=$(tail -n 1 $O/bench_stdout.txt | wc -c) lines=$(wc -l < $O/bench_stdout.txt)"
  timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/pytest_gpu_lean.log 2>&1; tail -n 2 $O/pytest_gpu_lean.log
  MBN_LAB=1 timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest_gpu_lab.log 2>&1; tail -n 2 $O/pytest_gpu_lab.log
  $M --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 --streams 2 --verify > $O/c_host_gpus1.txt 2>&1; tail -n 4 $O/c_host_gpus1.txt
  python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
else
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt --no-power --record $O/bench_under_rocprof_full.json > $O/bench_under_rocprof.txt 2> $O/bench_under_rocprof.err
  cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unf -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-fuse-stem --fuse-blocks 0 --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt --no-power --record $O/bench_unfused_under_rocprof_full.json > $O/bench_unfused_under_rocprof.txt 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_unf/*/*kernel_stats.csv | head -1) $O/kernel_stats_unfused.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -- python3 $R/bench.py --dtype bf16 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power --record $O/bench_bf16_under_rocprof_full.json > $O/bench_bf16_under_rocprof.txt 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_bf16/*/*kernel_stats.csv | head -1) $O/kernel_stats_bf16.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16s -- python3 $R/bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power --record $O/bench_bf16_05x160_under_rocprof_full.json > $O/bench_bf16_05x160_under_rocprof.txt 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_bf16s/*/*kernel_stats.csv | head -1) $O/kernel_stats_bf16_05x160.csv
  rm -rf $O/stats $O/stats_unf $O/stats_bf16 $O/stats_bf16s
  echo "stats done"; head -n 6 $O/kernel_stats.csv
fi
echo done

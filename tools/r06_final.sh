#!/bin/bash
# tools/r06_final.sh — the evidence set of round 6, ONE job on ONE box; outputs under gpurun_out/r06_final/
#   1. the default bench line exactly as the driver runs it (+ its side file)
#   2. rocprofv3 --kernel-trace --stats of the bench command (default fusion; one launch per layer; bf16 both sizes) + steady-state durations
#      (first 6 calls of every kernel dropped: tools/steady_stats.py)
#   3. GPU test suite on the shipped and on the lab library, C host with --verify, smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
M=$R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet
python3 bench.py --gpus 1 --steps 20 --warmup 5 --record $O/bench_full_record.json > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$? bytes=$(tail -n 1 $O/bench_stdout.txt | wc -c) lines=$(wc -l < $O/bench_stdout.txt)"
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
prof() {  # tag, bench args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt --no-power --record $O/bench_${tag}_under_rocprof_full.json "$@" > $O/bench_${tag}_under_rocprof.txt 2>> $O/bench_under_rocprof.err
  cp $(ls $O/stats_$tag/*/*kernel_stats.csv | head -1) $O/kernel_stats_$tag.csv
  python3 $R/tools/steady_stats.py $O/stats_$tag --drop 6 > $O/kernel_steady_$tag.csv
  rm -rf $O/stats_$tag
  echo "stats $tag done"
}
prof default --streams 1
prof unfused --streams 1 --no-fuse-stem --fuse-blocks 0
prof bf16 --dtype bf16 --batch 512
prof bf16_05x160 --dtype bf16 --alpha 0.5 --res 160 --batch 512
head -n 8 $O/kernel_steady_default.csv
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/pytest_gpu_lean.log 2>&1; tail -n 2 $O/pytest_gpu_lean.log
MBN_LAB=1 timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest_gpu_lab.log 2>&1; tail -n 2 $O/pytest_gpu_lab.log
$M --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 --streams 2 --verify > $O/c_host_gpus1.txt 2>&1; tail -n 4 $O/c_host_gpus1.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
echo done

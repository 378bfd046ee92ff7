#!/bin/bash
# round 5, step M: (1) bench line with package power, as the driver runs it; (2) bf16 layer 15 ablations with power / clock
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05m; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; tail -n 1 $O/bench_stdout.txt | wc -c
python3 tools/power_probe_bf16.py | tee $O/power_probe_bf16_L15.txt

#!/bin/bash
# round 6, step V: resident blocks in the net runner: bf16 tests, then the bf16 0.5x160 bench with and without
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06v; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 5 $O/pytest_bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 40 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resident', round(d['value']), d['ms_per_step'], d['stages_frac'])" | tee -a $O/bench_05x160.txt
  python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 40 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power --no-fuse-resident 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('per block', round(d['value']), d['ms_per_step'], d['stages_frac'])" | tee -a $O/bench_05x160.txt
done

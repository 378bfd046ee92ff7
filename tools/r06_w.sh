#!/bin/bash
# round 6, step W: resident kernel with the depthwise constants through LDS: parity, isolation time, ablation, net A/B
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06w; mkdir -p $O
timeout -k 10 300 python3 tools/res_bench.py | tee $O/res_bench.txt || exit 1
for bits in 1 2 4 8 16 3 11 15; do
  echo -n "dbg $bits: "; MBN_LAB=1 timeout -k 10 120 python3 tools/res_bench.py --tune exp0=$((900+bits)) | tail -n 1 || exit 1
done | tee $O/res_ablate.txt
for rep in 1 2; do
  python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 40 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resident', round(d['value']), d['ms_per_step'], d['stages_frac'])" | tee -a $O/bench_05x160.txt
  python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 40 --warmup 5 --no-cpu-baseline --no-unfused-stages --no-power --no-fuse-resident 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('per block', round(d['value']), d['ms_per_step'], d['stages_frac'])" | tee -a $O/bench_05x160.txt
done

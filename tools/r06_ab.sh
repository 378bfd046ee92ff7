#!/bin/bash
# round 6, step AB: hipGraph replay of the step on the short configurations (bf16 0.5x160 batch 512: 17 launches in 0.5 ms; fp32 batch 1)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06ab; mkdir -p $O
for rep in 1 2; do
  for g in "" "--graph"; do
    python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 60 --warmup 5 $g --no-cpu-baseline --no-unfused-stages --no-power --no-profile 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 0.5x160 b512 [$g]:', round(d['value']), d['ms_per_step'])" | tee -a $O/graph.txt || exit 1
    python3 bench.py --dtype f32 --batch 1 --steps 200 --warmup 20 $g --no-cpu-baseline --no-unfused-stages --no-power --no-profile --no-configs-alt --no-pw-emul-alt 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 b1 [$g]:', round(d['value']), d['ms_per_step'])" | tee -a $O/graph.txt || exit 1
  done
done

#!/bin/bash
# round 6, step AA: bf16 lines on one / two / three streams (round 2 measured +1.3 % at 1.0x224 and -1.5 % at 0.5x160 for two; kernels have changed since)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06aa; mkdir -p $O
for rep in 1 2; do
for st in 1 2 3; do
  for cfg in "1.0 224" "0.5 160"; do
    set -- $cfg
    python3 bench.py --dtype bf16 --alpha $1 --res $2 --batch 512 --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-unfused-stages --no-power --no-profile 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('alpha $1 res $2 streams $st:', round(d['value']), d['ms_per_step'], (d.get('parity_check') or {}).get('ok'))" | tee -a $O/bf16_streams.txt || exit 1
  done
done
done

#!/bin/bash
# round 6, step AD: two streams half a network apart. The streams' sub-batches start 2 layers apart (net_stagger = 2): the same kind of kernel meets itself. In free-running mode the
# dependency "stream 1 starts its step when stream 0 has passed layer S" keeps the lag at S layers: S around 13 puts one sub-batch's HBM-bound first half under the other's MFMA-bound second half.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06ad; mkdir -p $O
for rep in 1 2; do
for s in 2 6 10 12 14 16 20 24; do
    python3 bench.py --steps 30 --warmup 5 --streams 2 --tune net_stagger=$s --no-cpu-baseline --no-unfused-stages --no-power --no-profile --no-configs-alt --no-pw-emul-alt 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger $s:', round(d['value']), d['ms_per_step'])" | tee -a $O/stagger.txt || exit 1
done
done

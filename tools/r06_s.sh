#!/bin/bash
# round 6, step S: sub-batch streams with the new kernels: 1 / 2 / 3 streams (the bench's timed region, no per-kernel events)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06s; mkdir -p $O
for rep in 1 2; do for s in 1 2 3 4; do
  python3 bench.py --steps 40 --warmup 5 --streams $s --no-cpu-baseline --no-unfused-stages --no-configs-alt --no-pw-emul-alt --no-power --profile-steps 1 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $s', round(d['value']), d['ms_per_step'])" | tee -a $O/streams.txt
done; done

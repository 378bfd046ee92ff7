#!/bin/bash
# round 6, step AC: do the two streams' kernels co-reside? pw_gemm<64,64> at 4 workgroups per CU takes all 160 KB of LDS and 416 of 512 VGPRs per lane: no depthwise wave (115-128 VGPRs) fits beside it.
# At 3 per CU (lab knob misc = 3) one depthwise workgroup per CU does. Headline step on 1 / 2 streams with 4 / 3 / 2 GEMM workgroups per CU.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06ac; mkdir -p $O
for rep in 1 2; do
for st in 2 1; do
  for m in 0 3 2; do
    MBN_LAB=1 python3 bench.py --steps 30 --warmup 5 --streams $st --tune misc=$m --no-cpu-baseline --no-unfused-stages --no-power --no-profile --no-configs-alt --no-pw-emul-alt 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $st gemm wg/cu $m (0 = 4):', round(d['value']), d['ms_per_step'])" | tee -a $O/coresidency.txt || exit 1
  done
done
done

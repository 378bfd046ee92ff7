#!/bin/bash
# round 6, step D: what the MFMA-only skeleton of dwpw3 pays for (block 6-7 and 10-11): staircase of ablations on top of 7 = no x loads, no depthwise FMAs, no stores
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06d; mkdir -p $O
for v in 307 339 403 531 787 803; do
  echo "== dwpw_variant $v (bits $((v-300)))"; timeout -k 10 300 python3 tools/block_bench.py --blocks 6,10 --reps 20 --tune dwpw_variant=$v | tee -a $O/staircase.txt || exit 1
done

#!/bin/bash
# round 6, step K: compile-time ablations of dwpw3 (blocks 6-7, 10-11): no runtime switch inside the substep
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06k; mkdir -p $O
for v in 11 301 302 304 307 316 323 339 403 531 787 803 11; do
  echo "== dwpw_variant $v (bits $((v-300)))"; timeout -k 10 300 python3 tools/block_bench.py --blocks 6,10 --reps 20 --tune dwpw_variant=$v | tee -a $O/ablation_ct.txt || exit 1
done

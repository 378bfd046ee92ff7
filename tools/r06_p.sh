#!/bin/bash
# round 6, step P: dwpw3 with 8-byte B fragment reads (lab exp2 = 4) against the shipped form
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06p; mkdir -p $O
for b in 6 10; do echo "== block $b batch 64 exp2=4"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 64 --tune exp2=4 2>&1 | tee -a $O/debug.txt || exit 1; done
for rep in 1 2 3; do for t in "exp2=0" "exp2=4"; do
  echo "== dwpw3 $t"; timeout -k 10 300 python3 tools/block_bench.py --blocks 6,10 --reps 30 --tune dwpw_variant=11 --tune $t | tee -a $O/b64.txt || exit 1
done; done

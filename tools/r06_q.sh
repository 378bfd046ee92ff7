#!/bin/bash
# round 6, step Q: dwpw3, a tile's first substep with its depthwise part and window loads AHEAD of the previous tile's stores (lab exp2 = 3; 5 = with 8-byte B fragments)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06q; mkdir -p $O
for t in 3 5; do for b in 6 10; do echo "== block $b batch 64 exp2=$t"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 64 --tune exp2=$t 2>&1 | tee -a $O/debug.txt || exit 1; done; done
for rep in 1 2 3; do for t in "exp2=0" "exp2=3" "exp2=5"; do
  echo "== dwpw3 $t"; timeout -k 10 300 python3 tools/block_bench.py --blocks 6,10 --reps 30 --tune dwpw_variant=11 --tune $t | tee -a $O/sched3.txt || exit 1
done; done

#!/bin/bash
# round 6, step J: dwpw3 scheduling forms (lab exp2): 0 = four pinned groups per substep, 1 = the compiler's order, 2 = MFMA : 2 VALU : 1 memory interleave requested
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06j; mkdir -p $O
for t in "exp2=1" "exp2=2"; do for b in 6 10; do echo "== block $b batch 64 $t"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 64 --tune $t 2>&1 | tee -a $O/debug.txt || exit 1; done; done
for rep in 1 2; do for t in "exp2=0" "exp2=1" "exp2=2"; do
  echo "== dwpw3 $t"; timeout -k 10 300 python3 tools/block_bench.py --blocks 6,10 --reps 30 --tune dwpw_variant=11 --tune $t | tee -a $O/sched.txt || exit 1
done; done

#!/bin/bash
# round 6, step A: the wave-private fused block (mbn_f32_dwpw3.hip, lab dwpw_variant = 11) — parity, then A/B against the shipped dwpw2
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06a; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "test_f32_dwpw_fused and not emul" > $O/pytest_blocks.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 5 $O/pytest_blocks.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "== shipped rep $rep";  timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 | tee -a $O/block_shipped.txt || exit 1
  echo "== dwpw3 (variant 11) rep $rep"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=11 | tee -a $O/block_dwpw3.txt || exit 1
done

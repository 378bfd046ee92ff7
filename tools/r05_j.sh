#!/bin/bash
# round 5, step J: two 4-wave workgroups per CU on 64-row tiles (lab dwpw_variant = 10) against the shipped 8-wave workgroup
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
for rep in 1 2 3; do
  echo "== shipped (one 8-wave workgroup per CU, 128-row tiles) rep $rep" | tee -a $O/block_ab.txt;  python3 tools/block_bench.py --blocks 4,6 --reps 30 | grep "^L" | tee -a $O/block_ab.txt
  echo "== two 4-wave workgroups per CU, 64-row tiles (dwpw_variant=10) rep $rep" | tee -a $O/block_ab.txt; python3 tools/block_bench.py --blocks 4,6 --reps 30 --tune dwpw_variant=10 | grep "^L" | tee -a $O/block_ab.txt
done

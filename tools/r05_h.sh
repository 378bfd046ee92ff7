#!/bin/bash
# round 5, step H: stores of a finished tile spread under the next step's MFMA groups (default here) against the burst behind the barrier (exp0 = 52)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dwpw or fused_block or net_default" > $O/pytest_blocks.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_blocks.log
for rep in 1 2 3; do
  echo "== spread stores rep $rep" | tee -a $O/block_ab.txt;  python3 tools/block_bench.py --blocks 4,6 --reps 30 | grep "^L" | tee -a $O/block_ab.txt
  echo "== store burst (exp0=52) rep $rep" | tee -a $O/block_ab.txt; python3 tools/block_bench.py --blocks 4,6 --reps 30 --tune exp0=52 | grep "^L" | tee -a $O/block_ab.txt
done

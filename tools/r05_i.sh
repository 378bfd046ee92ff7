#!/bin/bash
# round 5, step I: the fused block kernel as shipped now (interleaved step, taps just in time, full-rate offsets, packed epilogue BN) against the burst form of rounds 2-4 (dwpw_variant = 9)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dwpw or fused_block or net_default" > $O/pytest_blocks.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_blocks.log
for rep in 1 2 3; do
  echo "== shipped rep $rep" | tee -a $O/block_ab.txt;  python3 tools/block_bench.py --blocks 4,6,8 --reps 30 | grep "^L" | tee -a $O/block_ab.txt
  echo "== burst form of rounds 2-4 (dwpw_variant=9) rep $rep" | tee -a $O/block_ab.txt; python3 tools/block_bench.py --blocks 4,6,8 --reps 30 --tune dwpw_variant=9 | grep "^L" | tee -a $O/block_ab.txt
done

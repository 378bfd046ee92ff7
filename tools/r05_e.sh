#!/bin/bash
# round 5, step E: fused block kernel with full-rate tile offsets + packed epilogue BN (shipped) against the general offsets (lab exp0 = 51) and the burst form (dwpw_variant = 9)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dwpw or fused_block or net_default or headline" > $O/pytest_blocks.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_blocks.log
for rep in 1 2 3; do
  echo "== shipped (IL + fast offsets + packed epilogue) rep $rep" | tee -a $O/block_ab.txt;  python3 tools/block_bench.py --blocks 4,6,8 --reps 30 | grep "^L" | tee -a $O/block_ab.txt
  echo "== general offsets (exp0=51) rep $rep" | tee -a $O/block_ab.txt; python3 tools/block_bench.py --blocks 4,6,8 --reps 30 --tune exp0=51 | grep "^L" | tee -a $O/block_ab.txt
  echo "== burst form of rounds 2-4 (dwpw_variant=9; general offsets) rep $rep" | tee -a $O/block_ab.txt; python3 tools/block_bench.py --blocks 4,6,8 --reps 30 --tune dwpw_variant=9 | grep "^L" | tee -a $O/block_ab.txt
done

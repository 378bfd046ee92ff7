#!/bin/bash
# round 5, step B: fused block kernel, interleaved step (IL, shipped now) against the burst form of rounds 2-4 (lab dwpw_variant = 9), alternating runs
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dwpw or fused_block or net_default" > $O/pytest_blocks.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_blocks.log
for rep in 1 2 3; do
  echo "== IL (shipped) rep $rep";  python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 | tee -a $O/block_il.txt
  echo "== burst (variant 9) rep $rep"; python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=9 | tee -a $O/block_burst.txt
done

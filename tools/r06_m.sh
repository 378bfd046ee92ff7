#!/bin/bash
# round 6, step M: block 12-13 (256 -> 512, stride 2, four slices) on dwpw3 against the two launches; blocks 4-11 after the register-pressure edits
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06m; mkdir -p $O
timeout -k 10 200 python3 tools/dwpw3_debug.py --block 12 --batch 256 2>&1 | tee -a $O/debug.txt || exit 1
for rep in 1 2; do
echo "== dwpw2 (variant 12)";  timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10,12 --reps 30 --tune dwpw_variant=12 | tee -a $O/block_dwpw2.txt || exit 1
echo "== dwpw3 (variant 11)"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10,12 --reps 30 --tune dwpw_variant=11 | tee -a $O/block_dwpw3.txt || exit 1
done
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "dwpw_fused" > $O/pytest_blocks.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_blocks.log

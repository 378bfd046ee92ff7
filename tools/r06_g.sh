#!/bin/bash
# round 6, step G: dwpw3 v2 (zigzag patches, 128-byte lines for stride 1 / Cin <= 128): parity, then A/B over the band height and the half-round width
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06g; mkdir -p $O
for b in 6 4 8 10; do echo "== block $b batch 256"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 256 2>&1 | tee -a $O/debug.txt || exit 1; done
for b in 6 8; do for t in "exp1=1" "exp1=2" "exp0=16" "exp0=32"; do echo "== block $b batch 37 $t"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 37 --tune $t 2>&1 | tee -a $O/debug.txt || exit 1; done; done
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "test_f32_dwpw_fused and not emul" > $O/pytest_blocks.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_blocks.log
[ $rc -ne 0 ] && exit $rc
echo "== shipped";  timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=12 | tee -a $O/block_shipped.txt || exit 1
for t in "" "--tune exp1=1" "--tune exp1=2" "--tune exp0=16" "--tune exp0=32"; do
  echo "== dwpw3 (variant 11) $t"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=11 $t | tee -a $O/block_dwpw3.txt || exit 1
done

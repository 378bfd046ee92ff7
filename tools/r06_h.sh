#!/bin/bash
# round 6, step H: dwpw3 with 32-channel half-rounds also for stride 2 (separable offsets): parity + A/B
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06h; mkdir -p $O
for b in 4 8; do echo "== block $b batch 256"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch 256 2>&1 | tee -a $O/debug.txt || exit 1; done
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "test_f32_dwpw_fused and not emul" > $O/pytest_blocks.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_blocks.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
echo "== shipped";  timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=12 | tee -a $O/block_shipped.txt || exit 1
echo "== dwpw3 (variant 11)"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 30 --tune dwpw_variant=11 | tee -a $O/block_dwpw3.txt || exit 1
done
for v in 301 302 304 316 307; do
  echo "== dwpw_variant $v"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 20 --tune dwpw_variant=$v | tee -a $O/ablation.txt || exit 1
done

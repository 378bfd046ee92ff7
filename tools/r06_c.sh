#!/bin/bash
# round 6, step C: dwpw3 after the store-data hazard fix — full-tensor comparison at three batches, then the ablation matrix
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06c; mkdir -p $O
for b in 6 4 8 10; do for n in 64 256; do echo "== block $b batch $n"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch $n 2>&1 | tee -a $O/debug.txt || exit 1; done; done
for v in 11 301 302 304 316 305 307 323; do
  echo "== dwpw_variant $v"; timeout -k 10 300 python3 tools/block_bench.py --blocks 4,6,8,10 --reps 20 --tune dwpw_variant=$v | tee -a $O/ablation.txt || exit 1
done

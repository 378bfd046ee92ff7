#!/bin/bash
# round 5, step C: ablation matrix of the fused block kernel (burst form, DBG build: dwpw_variant = 100 + bits; 1 no x loads, 2 no depthwise math, 4 no stores, 8 no filter DMA, 16 no MFMA)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
for v in 9 100 116 105 113 117 102 104 101 108 121 125 127; do
  echo "== dwpw_variant=$v" | tee -a $O/ablation.txt
  python3 tools/block_bench.py --blocks 4,6,8 --reps 20 --tune dwpw_variant=$v 2>&1 | grep "^L" | tee -a $O/ablation.txt
done

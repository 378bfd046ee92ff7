#!/bin/bash
# round 5, step F: what in the non-MFMA "skeleton" of the fused block kernel adds to the MFMA time? burst-form DBG build, block 6-7, dwpw_variant = 100 + bits
# bits: 1 no x loads, 2 no depthwise math, 4 no stores, 8 no filter DMA, 16 no MFMA, 1024 no tap reads, 2048 one fragment read per step, 4096 no barrier, 8192 no accumulator zeroing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05f; mkdir -p $O
run() { echo "== bits $1 ($2)" | tee -a $O/skeleton.txt; python3 tools/block_bench.py --blocks 6 --reps 20 --tune dwpw_variant=$((100 + $1)) 2>&1 | grep "^L" | tee -a $O/skeleton.txt; }
run 0 "full"
run 13 "no HBM traffic: no x loads, no stores, no filter DMA"
run 15 "... and no depthwise math"
run 1039 "... and no tap reads"
run 3087 "... and one fragment read per step"
run 7183 "... and no barrier"
run 15375 "... and no accumulator zeroing: MFMAs + bookkeeping + set_offsets + epilogue arithmetic off"
run 2063 "no HBM, no dw math, one fragment read per step (tap reads on)"
run 4111 "no HBM, no dw math, no barrier"
run 2048 "full but one fragment read per step"
run 1024 "full but no tap reads"
run 4096 "full but no barrier"

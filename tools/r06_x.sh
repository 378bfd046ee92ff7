#!/bin/bash
# round 6, step X: bf16 depthwise on the 14x14 / 7x7 layers at batch 512: row segments and 4-channel lanes (latency-bound? 41 us for 206 MB where fp32 takes 31 us for 410 MB)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06x; mkdir -p $O
timeout -k 10 300 python3 tools/layer_bench.py --dtype bf16 --batch 512 --layers 12,14,24,26 --iters 30 --tune dw_nseg=0,2,3,4,7 | tee $O/bf16_dw_nseg.txt || exit 1
timeout -k 10 300 python3 tools/layer_bench.py --dtype bf16 --batch 512 --layers 12,14,24,26 --iters 30 --tune dw_variant=0,32,1 | tee $O/bf16_dw_variant.txt || exit 1
timeout -k 10 300 python3 tools/layer_bench.py --dtype f32 --batch 512 --layers 14 --iters 30 | tee $O/f32_dw_b512.txt

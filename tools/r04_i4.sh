#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
for B in 16 32 64 128 256 512; do echo "## batch $B"; python tools/layer_bench.py --batch $B --layers 4 --iters 60 --warmup 5 --tune dw_nseg=0,28,56; done
echo "## res 160 (80 -> 40 rows), alpha 0.5, batch 512"; python tools/layer_bench.py --batch 512 --alpha 0.5 --res 160 --layers 4,8 --iters 60 --warmup 5 --tune dw_nseg=0,20,40
echo "## res 320 alpha 1 batch 128 (160 -> 80 rows)"; python tools/layer_bench.py --batch 128 --res 320 --layers 4,8 --iters 40 --warmup 5 --tune dw_nseg=0,40,80
} > $O/r04i_dw_l4_batches.txt 2>&1
grep "^L\|##" $O/r04i_dw_l4_batches.txt

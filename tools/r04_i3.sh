#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
for L in 2 4 6 8 10 12; do python tools/layer_bench.py --layers $L --iters 60 --warmup 5 --tune exp0=0,1 --tune dw_nseg=0,8,14,16,28,56; done
} > $O/r04i_dw_march_nseg.txt 2>&1
cat $O/r04i_dw_march_nseg.txt

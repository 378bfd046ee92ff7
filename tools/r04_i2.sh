#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python tools/layer_bench.py --layers 4 --iters 60 --warmup 5 --tune exp0=0,6,16 --tune dw_nseg=0,4,5,6,7,8,10,14,28 > $O/r04i_dw_l4.txt 2>&1
python tools/layer_bench.py --layers 8 --iters 60 --warmup 5 --tune exp0=0,6,16 --tune dw_nseg=0,2,4,7,14 >> $O/r04i_dw_l4.txt 2>&1
sort -k6 -n $O/r04i_dw_l4.txt | head -40

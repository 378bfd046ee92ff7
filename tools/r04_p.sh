#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
echo "#### fp32 pw_gemm, lab pw_tile: 3 = shipped 64x64 (4 waves of 32x32, 4 WG/CU), 9 = 64x128 and 10 = 128x64 with 8 waves of 32x32 (2 WG/CU: same waves per CU, same rounds at M = 50176, 25 % fewer staged bytes per flop), 5 = 128x128 / 32x64 waves"
python tools/layer_bench.py --custom-pw "49152,512,512;50176,512,512;12544,1024,1024;50176,256,512" --iters 200 --warmup 100 --tune pw_tile=3,9,10,5,3,9,10,5
} > $O/r04p_gemm_tiles.txt 2>&1
cat $O/r04p_gemm_tiles.txt

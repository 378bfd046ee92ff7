#!/bin/bash
# round 4, call A: new tests, fp32 GEMM ablation table, the counter passes that aborted in round 3 (item 6)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "self_launch or pool_fc or tail_one or dist_single or multi_rank" > $O/r04a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04a_pytest.log
tail -3 $O/r04a_pytest.log
{
echo "#### fp32 pw_gemm ablations (lab, exp1 = ABL bits: 1 no LDS-DMA in the k-loop, 2 no barriers in the k-loop, 4 no epilogue stores, 8 no LDS fragment reads in the k-loop);"
echo "#### pw_tile 3 = 64x64 (4 waves of 32x32, 4 WG/CU, shipped), 5 = 128x128 (8 waves of 32x64, 2 WG/CU), 1 = 128x128 (4 waves of 64x64, 2 WG/CU); M = 49152 = whole rounds of every form"
for t in 3 5 1; do
  python tools/layer_bench.py --custom-pw 49152,512,512 --iters 300 --warmup 150 --tune pw_tile=$t --tune exp1=0,1,2,4,8,3,7,9,10,12,15
done
} > $O/r04a_gemm_ablation.txt 2>&1
tail -5 $O/r04a_gemm_ablation.txt
tools/r04_pmc_diag.sh bf16L15 -- --layers 15 --dtype bf16 --batch 512 --iters 20 --warmup 3
PMC_TARGET=tools/block_bench.py tools/r04_pmc_diag.sh dwpw2b6 -- --blocks 6 --reps 10
ls $O | grep pmc_ | head -50
cat $O/pmc_bf16L15_summary.txt | head -60

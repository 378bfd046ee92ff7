#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06u; mkdir -p $O
for v in 900 901 902 904 908 916 903 907 915 931; do echo "== exp0=$v (bits $((v-900)))"; MBN_LAB=1 timeout -k 10 200 python3 tools/res_bench.py --reps 20 --tune exp0=$v | tee -a $O/res_ablation.txt; done

#!/bin/bash
# round 4, call C: clock held by the fp32 GEMM and its ablations, MFMA-vs-fmaf-chain bits, counter passes over the whole fp32 step, bf16 tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python tools/mfma_vs_fma_chain.py > $O/r04c_mfma_vs_chain.txt 2>&1; cat $O/r04c_mfma_vs_chain.txt
python tools/gemm_clock_ablation.py > $O/r04c_gemm_clock_ablation.txt 2>&1; cat $O/r04c_gemm_clock_ablation.txt
python -m pytest tests -m gpu -x -q -k "bf16" > $O/r04c_pytest_lean.log 2>&1; echo "rc=$?" >> $O/r04c_pytest_lean.log; tail -3 $O/r04c_pytest_lean.log
MBN_LAB=1 python -m pytest tests -m gpu -x -q -k "bf16_dwpw_fused or k256" > $O/r04c_pytest_lab.log 2>&1; echo "rc=$?" >> $O/r04c_pytest_lab.log; tail -3 $O/r04c_pytest_lab.log
export PMC_TARGET=bench.py MBN_LAB=0
tools/r04_pmc_diag.sh step -- --steps 6 --warmup 2 --no-cpu-baseline --no-configs-alt --no-pw-emul-alt --streams 1
grep -A 40 "stem_fused_f32" $O/pmc_step_summary.txt | head -45

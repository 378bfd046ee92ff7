#!/bin/bash
# round 6, step L: stem with the conv1 output rows padded to 40 floats (LDS bank conflicts): parity tests, A/B, the counter pair
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06l; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "stem" > $O/pytest_stem.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_stem.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do timeout -k 10 200 python3 tools/stem_bench.py --variants 0,6 | tee -a $O/stem_ab.txt || exit 1; done
export PMC_TARGET=tools/stem_bench.py PMC_TIMEOUT=200
tools/pmc_pass.sh r06l_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -- --reps 5 --variants 0,6 || echo "pmc rc=$?"
python3 tools/pmc_diag_summary.py gpurun_out/pmc_r06l_lds > $O/pmc_stem_lds.txt 2>&1; grep -A 12 "stem_fused" $O/pmc_stem_lds.txt

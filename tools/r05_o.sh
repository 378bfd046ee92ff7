#!/bin/bash
# round 5, step O: instruction counts per kernel of the fp32 step (SQ_INSTS_{VALU,SALU,LDS,VMEM} / SQ_INSTS_MFMA before / after), two PMC passes over bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PMC_TARGET=bench.py PMC_TIMEOUT=280
A="--steps 6 --warmup 2 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --streams 1"
bash tools/pmc_pass.sh r05o_sq3 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR -- $A; echo "pass sq3 rc=$?"
bash tools/pmc_pass.sh r05o_sq2 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL -- $A; echo "pass sq2 rc=$?"
bash tools/pmc_pass.sh r05o_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -- $A; echo "pass sq1 rc=$?"
python3 tools/pmc_diag_summary.py gpurun_out/pmc_r05o_* > gpurun_out/pmc_r05o_summary.txt 2>&1
rm -rf gpurun_out/pmc_r05o_sq1 gpurun_out/pmc_r05o_sq2 gpurun_out/pmc_r05o_sq3
wc -l gpurun_out/pmc_r05o_summary.txt

#!/bin/bash
# round 6, step E: counter passes over block 6-7 and 4-5, dwpw3 (lab dwpw_variant = 11) against the shipped dwpw2 (12)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=gpurun_out/r06e; mkdir -p $O
export PMC_TARGET=tools/block_bench.py PMC_TIMEOUT=200
P=$R/tools/pmc_pass.sh
for v in 11 12; do
  tag=r06e_v$v
  A="--blocks 4,6 --reps 6 --tune dwpw_variant=$v"
  $P ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -- $A || echo "pass sq1 rc=$?"
  $P ${tag}_sq2 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL -- $A || echo "pass sq2 rc=$?"
  $P ${tag}_sq3 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR -- $A || echo "pass sq3 rc=$?"
  $P ${tag}_ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum -- $A || echo "pass ta1 rc=$?"
  $P ${tag}_tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum -- $A || echo "pass tcp2 rc=$?"
  $P ${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -- $A || echo "pass tcc rc=$?"
  python3 tools/pmc_diag_summary.py gpurun_out/pmc_${tag}_* > $O/pmc_v${v}_summary.txt 2>&1
  grep -A 40 "dwpw" $O/pmc_v${v}_summary.txt | head -120
done

#!/bin/bash
# tools/r03_pmc_diag.sh <tag> -- <layer_bench args>
# Where does a kernel's time go? Five rocprofv3 --pmc passes (counters in their own runs, kernel trace only) over one
# tools/layer_bench.py invocation: wave-cycle buckets + matrix-pipe busy, VMEM/LDS issue, TA/TCP stalls, L1->L2 latency, L2 hit rate.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift; shift
P=$R/tools/pmc_pass.sh
$P ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -- "$@" &&
$P ${tag}_sq2 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL -- "$@" &&
$P ${tag}_ta TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -- "$@" &&
$P ${tag}_tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -- "$@" &&
$P ${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -- "$@"
python3 $R/tools/pmc_diag_summary.py $R/gpurun_out/pmc_${tag}_* > $R/gpurun_out/pmc_${tag}_summary.txt 2>&1

#!/bin/bash
# tools/r04_pmc_diag.sh <tag> -- <layer_bench args>
# Where does a kernel's time go? rocprofv3 --pmc passes (counters in their own runs, kernel trace only) over one
# tools/layer_bench.py invocation: wave-cycle buckets + matrix-pipe busy, VMEM/LDS issue, TA stalls, TCP (L1) stalls and L1->L2 latency, L2 hit rate.
# Round 3's script asked for 4 TA_* + GRBM and 5 TCP_* counters in one pass each; that over-subscribes the per-block counter slots and rocprofv3
# aborts with error 38 (profiles/LOG.md R3.1 recorded it, wrongly, as a property of the pool). Here: at most 2 counters of a texture block per pass,
# GRBM_GUI_ACTIVE with the SQ counters. A pass that still does not fit fails fast (pmc_pass.sh) and the remaining passes go on.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift; shift
P=$R/tools/pmc_pass.sh
$P ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -- "$@"
$P ${tag}_sq2 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL -- "$@"
$P ${tag}_sq3 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR -- "$@"
$P ${tag}_ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum -- "$@"
$P ${tag}_ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -- "$@"
$P ${tag}_tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum -- "$@"
$P ${tag}_tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum -- "$@"
$P ${tag}_tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum -- "$@"
$P ${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -- "$@"
python3 $R/tools/pmc_diag_summary.py $R/gpurun_out/pmc_${tag}_* > $R/gpurun_out/pmc_${tag}_summary.txt 2>&1

#!/bin/bash
# PMC passes over the fused compatibility + softmax kernel (run on the GPU box from the repo root):
#   bash tools/compat_pmc.sh  -> gpurun_out/cpmc_<n>/...counter_collection.csv, summarised by tools/compat_pmc.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
n=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
         "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_BUSY TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/cpmc_$n -- python3 $R/tools/compat_time.py > $R/gpurun_out/cpmc_$n.log 2>&1 || { tail -5 $R/gpurun_out/cpmc_$n.log; exit 1; }
done
python3 $R/tools/compat_pmc.py

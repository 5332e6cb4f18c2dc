#!/bin/bash
# PMC passes over the fused compatibility + softmax kernel (run on the GPU box from the repo root):
#   bash tools/compat_pmc.sh  -> gpurun_out/cpmc_<n>/...counter_collection.csv, summarised by tools/compat_pmc.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
n=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/cpmc_$n -- python3 $R/tools/compat_time.py > $R/gpurun_out/cpmc_$n.log 2>&1 || { tail -5 $R/gpurun_out/cpmc_$n.log; exit 1; }
done
python3 $R/tools/compat_pmc.py

#!/bin/bash
# PMC passes over the lattice build's kernels (what bounds k_chunk_masks / k_insert / k_elevate / k_final_vid / k_neighbors?).
# Run on the GPU box from the repo root:  bash tools/build_pmc.sh  -> gpurun_out/bpmc_p<n>/ ; summarised into
# gpurun_out/build_pmc.json by tools/build_pmc.py (separate --pmc passes, never combined with other trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bpmc_stats -- python3 $R/tools/reftable_time.py c3 > $R/gpurun_out/bpmc_stats.log 2>&1 || { echo "stats pass failed"; tail -3 $R/gpurun_out/bpmc_stats.log; exit 1; }
n=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/bpmc_p$n -- python3 $R/tools/reftable_time.py c3 > $R/gpurun_out/bpmc_p$n.log 2>&1 || { echo "pmc pass '$c' failed"; tail -3 $R/gpurun_out/bpmc_p$n.log; }
  echo "pass $n done"
done
cd $R && python3 tools/build_pmc.py

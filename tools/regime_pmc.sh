#!/bin/bash
# Kernel stats + PMC passes of the filter's kernels under feature regimes other than the default (what
# bounds the natural-image regimes?).  Run on the GPU box from the repo root:  bash tools/regime_pmc.sh [workload]
# -> gpurun_out/rpmc_<regime>_<pass>/ ; summarised into gpurun_out/regime_pmc.json by tools/regime_pmc.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp
Q="--workload $WL --no-cpu-baseline --no-regimes --no-mean-field --no-small-image"
for reg in "default:" "tsu_1_1:--tsukuba 0.1,0.1" "tsu_08_03:--tsukuba 0.08,0.03"; do
  tag=${reg%%:*}; opt=${reg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rpmc_${tag}_stats -- python3 $R/bench.py $Q $opt --steps 20 --warmup 5 > $R/gpurun_out/rpmc_${tag}_stats.log 2>&1 || { echo "stats pass failed for $tag"; tail -3 $R/gpurun_out/rpmc_${tag}_stats.log; exit 1; }
  n=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
    n=$((n+1))
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/rpmc_${tag}_p$n -- python3 $R/bench.py $Q $opt --steps 3 --warmup 1 > $R/gpurun_out/rpmc_${tag}_p$n.log 2>&1 || { echo "pmc pass '$c' failed for $tag"; tail -3 $R/gpurun_out/rpmc_${tag}_p$n.log; }
  done
  echo "regime $tag done"
done
cd $R && python3 tools/regime_pmc.py $WL

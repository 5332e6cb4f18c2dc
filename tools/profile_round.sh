#!/bin/bash
# Run on the GPU box (tools/gpu.sh -- 'bash tools/profile_round.sh TAG'): the default bench, the same command
# under rocprofv3 kernel stats, three separate PMC passes (never combined with other trace domains), and the
# exact-arithmetic bench.  Condense afterwards with tools/save_profiles.py TAG prof_TAG prof_TAG_exact pmc_TAG.
set -o pipefail
tag=${1:-r01_final}
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_final.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_final_prof.log 2>&1 || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_exact -- python3 $R/bench.py --no-cpu-baseline --exact > $R/gpurun_out/bench_final_exact.log 2>&1 || exit 1
grep "^{" $R/gpurun_out/bench_final.log | cut -c1-200

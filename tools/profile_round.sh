#!/bin/bash
# Run on the GPU box (tools/gpu.sh -- 'bash tools/profile_round.sh TAG'): the default bench, the same step under
# rocprofv3 kernel stats, three separate PMC passes (never combined with other trace domains), the exact-arithmetic
# bench, the mean-field iteration (k_compat_softmax) under kernel stats + PMC, and the regime sweep.
# Condense afterwards with tools/save_profiles.py TAG prof_TAG prof_TAG_exact pmc_TAG.
set -o pipefail
tag=${1:-r01_final}
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_final.log 2>$R/gpurun_out/bench_final.err || exit 1
cd /tmp && export TMPDIR=/tmp
# the profiled command is the headline step only: the regime sweep and the mean-field extras launch the same kernels
# on other lattices / other shapes and would blur the per-kernel averages
Q="--no-cpu-baseline --no-regimes --no-mean-field"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py $Q > $R/gpurun_out/bench_final_prof.log 2>&1 || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_exact -- python3 $R/bench.py $Q --exact > $R/gpurun_out/bench_final_exact.log 2>&1 || exit 1
# mean-field iteration: filter - Q, then the fused compatibility product + softmax (k_compat_softmax)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_mf -- python3 $R/bench.py --no-cpu-baseline --no-regimes --no-small-image --steps 5 > $R/gpurun_out/bench_final_mf.log 2>&1 || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_mf_$c -- python3 $R/bench.py --no-cpu-baseline --no-regimes --no-small-image --steps 2 --warmup 1 > $R/gpurun_out/pmc_${tag}_mf_$c.log 2>&1 || exit 1
done
cd $R && python tools/regimes.py --workloads c3,c2 --out gpurun_out/regimes_$tag.json > gpurun_out/regimes_$tag.log 2>&1 || exit 1
grep "^{" $R/gpurun_out/bench_final.log | cut -c1-200

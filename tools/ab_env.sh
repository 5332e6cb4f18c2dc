#!/bin/bash
# A/B of environment settings over feature regimes: tools/ab_env.sh "REGIMES" "ENV=a" "ENV=b" ...  (one regimes.py run per setting)
regs=$1; shift
mkdir -p gpurun_out
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =/' '___')
  env $cfg python tools/regimes.py --workloads ${WL:-c3} --regimes $regs --no-gather --out gpurun_out/ab_$tag.json 2>/dev/null | awk -v c="$cfg" '{print "[" c "] " $0}'
done

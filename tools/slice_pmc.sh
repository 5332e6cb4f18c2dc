#!/bin/bash
# LDS / wait counters of k_slice_tiled under two feature regimes (why is the slice 7 % slower on natural-image features?)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for reg in "" "--tsukuba 0.1,0.1"; do
  tag=$(echo "x$reg" | tr ' ,.-' '____')
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/spmc_${tag}_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-regimes --no-mean-field $reg > /dev/null 2>&1 || echo "pass $c failed"
  done
done
cd $R && python - <<'PY'
import csv, glob, collections
for tag in sorted(set(p.split('/')[1].split('_SQ')[0].split('_TCC')[0] for p in glob.glob('gpurun_out/spmc_*'))):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/{tag}_*/*/*counter_collection.csv'):
        for row in csv.DictReader(open(f)):
            if 'k_slice_tiled' in row['Kernel_Name']:
                agg[row['Counter_Name']].append(float(row['Counter_Value']))
    print(tag, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

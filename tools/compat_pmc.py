"""Summarise tools/compat_pmc.sh: per-dispatch means of every counter for k_compat_softmax (softmax epilogue)."""
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/cpmc_*/')):
    for f in glob.glob(d + '*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'k_compat_softmax<8, false, false>' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(f"{k:36s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")

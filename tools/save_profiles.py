"""Condense rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402  (hot_source_sha)

tag, stats_dir, exact_dir, pmc_prefix = sys.argv[1:5]


def newest(pattern):
    """gpurun_out/ keeps the files of earlier calls with the same tag: take the latest run's"""
    return max(glob.glob(pattern), key=os.path.getmtime)


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0] if not n.startswith('at::') else n.split('<')[0]


def stats(dirn, out):
    f = newest(f'gpurun_out/{dirn}/*/*kernel_stats.csv')
    rows = list(csv.DictReader(open(f)))
    with open(out, 'w') as fo:
        w = csv.writer(fo)
        w.writerow(['Kernel', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'], r['StdDev']])
    return rows


for x in stats(stats_dir, f'profiles/{tag}_bench_c3_kernel_stats.csv')[:5]:
    print(short(x['Name']), x['Calls'], round(float(x['AverageNs']) / 1e3, 1), 'us')
for x in stats(exact_dir, f'profiles/{tag}_bench_c3_exact_kernel_stats.csv')[:3]:
    print('exact', short(x['Name']), x['Calls'], round(float(x['AverageNs']) / 1e3, 1), 'us')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum'):
    f = newest(f'gpurun_out/{pmc_prefix}_{d}/*/*counter_collection.csv')
    for row in csv.DictReader(open(f)):
        n = short(row['Kernel_Name'])
        if n.startswith('k_'):
            agg[n][row['Counter_Name']].append(float(row['Counter_Value']))
pm = {}
for k, v in agg.items():
    if not any(t in k for t in ('splat', 'blur', 'slice')):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    fk, wk = m.get('FETCH_SIZE', 0), m.get('WRITE_SIZE', 0)
    pm[k] = {'dispatches_sampled': len(v['FETCH_SIZE']), 'FETCH_SIZE_KB_raw': round(fk, 1), 'WRITE_SIZE_KB': round(wk, 1),
             'fetch_bytes_corrected_x2': int(fk * 2048), 'write_bytes': int(wk * 1024), 'hbm_bytes_per_launch': int(fk * 2048 + wk * 1024),
             'L2_hit_rate': round(m.get('TCC_HIT_sum', 0) / max(1, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0)), 3)}
json.dump({'note': 'rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), bench.py c3, per-dispatch means. '
           'FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (it tallies 128-B requests at 64 B); WRITE_SIZE is exact.',
           'hot_kernel_source_sha': bench.hot_source_sha(),       # bench.py flags the figures as stale when the kernels change
           'kernels': pm}, open(f'profiles/{tag}_pmc_traffic.json', 'w'), indent=1)
for k, v in pm.items():
    print(k, v['hbm_bytes_per_launch'], v['L2_hit_rate'])
shutil.copy('gpurun_out/bench_final.log', f'profiles/{tag}_bench_c3.log')
shutil.copy('gpurun_out/bench_final_prof.log', f'profiles/{tag}_bench_c3_under_rocprof.log')
open(f'profiles/{tag}_bench_c3_exact.log', 'w').write(''.join(l for l in open('gpurun_out/bench_final_exact.log') if l.startswith('{')))

# mean-field iteration (k_compat_softmax): kernel stats + HBM traffic of the fused compatibility kernel
import os
if glob.glob(f'gpurun_out/prof_{tag}_mf/*/*kernel_stats.csv'):
    for x in stats(f'prof_{tag}_mf', f'profiles/{tag}_meanfield_kernel_stats.csv')[:4]:
        print('mean-field', short(x['Name']), x['Calls'], round(float(x['AverageNs']) / 1e3, 1), 'us')
    mf = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in [newest(f'gpurun_out/pmc_{tag}_mf_{d}/*/*counter_collection.csv')]:
            for row in csv.DictReader(open(f)):
                n = short(row['Kernel_Name'])
                if n.startswith('k_compat'):
                    mf[n][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {}
    for k, v in mf.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        fk, wk = m.get('FETCH_SIZE', 0), m.get('WRITE_SIZE', 0)
        out[k] = {'dispatches_sampled': len(v['FETCH_SIZE']), 'fetch_bytes_corrected_x2': int(fk * 2048), 'write_bytes': int(wk * 1024),
                  'hbm_bytes_per_launch': int(fk * 2048 + wk * 1024)}
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py c3 mean-field extras; FETCH_SIZE doubled (gfx950)',
               'kernels': out}, open(f'profiles/{tag}_meanfield_pmc_traffic.json', 'w'), indent=1)
    open(f'profiles/{tag}_bench_c3_mean_field.log', 'w').write(''.join(l for l in open('gpurun_out/bench_final_mf.log') if l.startswith('{')))
if os.path.exists(f'gpurun_out/regimes_{tag}.json'):
    shutil.copy(f'gpurun_out/regimes_{tag}.json', f'profiles/{tag}_regimes.json')

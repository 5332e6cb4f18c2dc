"""One mean-field iteration at C3 under `rocprofv3 --kernel-trace` and the analysis of its timeline.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mf_trace -- python3 tools/mf_trace.py
  python tools/mf_trace.py --analyze gpurun_out/mf_trace     kernels of the last iterations: start, duration, gap to the previous end
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == '--analyze':
    f = sorted(glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    # last occurrence of the compatibility kernel ends an iteration; print the two iterations before it
    idx = [i for i, n in enumerate(names) if 'k_compat' in n]
    lo, hi = idx[-3] + 1, idx[-1] + 1
    t0 = int(rows[lo]['Start_Timestamp'])
    prev_end = None
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        import re
        m = re.search(r'(k_\w+|Cijk\w+|\w+)(<[^(]*>)?\(', r['Kernel_Name'].replace('(anonymous namespace)::', ''))
        nm = (m.group(1) + (m.group(2) or '') if m else r['Kernel_Name'])[:48]
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  q{r.get("Queue_Id", "?")}  {nm}')
        prev_end = max(prev_end or e, e)
    print('two iterations:', (int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3, 'us')
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import torch, bench, phl
from crf.crf_module import charbonneir, compatibility_matrix, mean_field_step
H, W, L = 1536, 2048, 256
dev = torch.device('cuda')
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev))
g = torch.Generator(device=dev).manual_seed(5)
E0 = torch.rand((H * W, L), device=dev, generator=g) * 10
Q = torch.softmax(-E0, dim=1)
Qn = torch.empty_like(Q)
Mu = compatibility_matrix(lambda a, b: charbonneir(a, b, 3.0), torch.arange(L, dtype=torch.float32, device=dev))
Wf = lambda U: lat.filter(U, subtract_input=True)
for _ in range(12):
    mean_field_step(E0, Wf, Mu, Q, out=Qn)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    mean_field_step(E0, Wf, Mu, Q, out=Qn)
b.record(); torch.cuda.synchronize()
print('iteration ms', a.elapsed_time(b) / 10)

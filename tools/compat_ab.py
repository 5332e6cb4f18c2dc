"""Same-box A/B of library builds on the split compatibility kernel: python tools/compat_ab.py lib1.so lib2.so ...
(each in its own child process; three rounds, interleaved; prints the minimum and median of the steady-state time)."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, 'depth-estimation_amd')); sys.path.insert(0, %r)
import phl
n, L = 1536 * 2048, 256
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
E0 = torch.rand((n, L), device=dev, generator=g) * 10
X = torch.rand((n, L), device=dev, generator=g)
Mu = torch.rand((L, L), device=dev, generator=g) * 3
out = torch.empty_like(E0)
arith = os.environ.get('AB_ARITH', 'split')
f = lambda: phl.compat_softmax(E0, X, Mu, out=out, arith=arith)
for _ in range(15): f()
torch.cuda.synchronize()
ts = []
for r in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 10)
print(min(ts))
''' % (ROOT, ROOT)
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != 'default':
            env['PHL_LIB'] = os.path.abspath(l)
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=120)
        try:
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(l, 'FAILED', out.stderr[-400:])
for l in libs:
    if res[l]:
        print(f'{os.path.basename(l):28s} min {min(res[l]):.3f}  median {statistics.median(res[l]):.3f} ms', flush=True)

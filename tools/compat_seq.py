"""Per-launch times of phl.compat_softmax in a row of 24 back-to-back launches (clock ramp-up from idle), of isolated
launches 50 ms apart, and of rocBLAS mm in a row -- C3-size operands.  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import phl
n, L = 1536 * 2048, 256
g = torch.Generator(device='cuda').manual_seed(0)
E0 = torch.rand((n, L), device='cuda', generator=g) * 10
X = torch.rand((n, L), device='cuda', generator=g)
Mu = torch.rand((L, L), device='cuda', generator=g) * 3
out = torch.empty_like(E0)
phl.compat_softmax(E0, X, Mu, out=out); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
ev[0].record()
for k in range(24):
    phl.compat_softmax(E0, X, Mu, out=out)
    ev[k + 1].record()
torch.cuda.synchronize()
print('back-to-back per-launch ms:', ' '.join('%.2f' % ev[k].elapsed_time(ev[k + 1]) for k in range(24)))
import time
ts = []
for k in range(8):
    time.sleep(0.05)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); phl.compat_softmax(E0, X, Mu, out=out); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print('isolated launches ms:', ' '.join('%.2f' % t for t in ts))
G = torch.empty_like(E0)
ev[0].record()
for k in range(12):
    torch.mm(X, Mu, out=G); ev[k + 1].record()
torch.cuda.synchronize()
print('rocBLAS mm back-to-back ms:', ' '.join('%.2f' % ev[k].elapsed_time(ev[k + 1]) for k in range(12)))

import os, sys, time
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import torch, bench, phl
H, W = 1110, 1390
dev = torch.device('cuda')
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev))
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for vd in (int(a) for a in (sys.argv[1:] or (9, 10, 12, 13, 17, 20, 30, 50))):
    x = torch.rand((H * W, vd), device=dev)
    out = torch.empty_like(x)
    a = t(lambda: lat.filter(x, out=out))
    b = t(lambda: lat.filter(x, out=out, no_tiles=True))
    print(f'vd={vd:3d}: default {a:.3f} ms   gather kernels {b:.3f} ms', flush=True)

"""Robustness probe at 4x the benchmark's pixel count (12.6 M pixels): build, filter, properties."""
import sys, time
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import torch, phl, bench
H, W, L = 3072, 4096, 64
feat = bench.synthetic_features(H, W)
dev = torch.device('cuda')
t0 = time.time(); lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev)); torch.cuda.synchronize()
print('n', H * W, 'M', lat.M, 'M/n', lat.M / (H * W), 'build s', time.time() - t0, 'dev MB', lat.device_bytes / 2**20, lat.tile_stats(L))
x = torch.rand((H * W, L), device=dev)
y = lat.filter(x); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): y = lat.filter(x)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print('filter ms', dt * 1e3, 'Mpl/s', H * W * L / dt / 1e6)
ye = lat.filter(x, exact=True)
print('default vs exact max rel', float(((y - ye).abs() / ye.abs().clamp_min(1e-3 * float(ye.max()))).max()))
print('finite', bool(torch.isfinite(y).all()), 'min', float(y.min()))

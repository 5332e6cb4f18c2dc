"""Robustness probe at the top of the supported range: n (d+1) just under 2^30 candidates (the library's limit), few channels.
Default arithmetic against the exact mode (separate kernels and index structures), and channel independence."""
import sys, time
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import numpy as np, torch, phl, bench

side = int(sys.argv[1]) if len(sys.argv) > 1 else 13056          # 13056^2 * 6 = 1.0227e9 < 2^30
L = 4
t0 = time.time()
feat = bench.synthetic_features(side, side).reshape(-1, 5)
print('features', feat.shape, 'in', round(time.time() - t0, 1), 's', flush=True)
dev = torch.device('cuda')
ref = torch.from_numpy(feat).to(dev)
del feat
n = ref.shape[0]
t0 = time.time(); lat = phl.Lattice(ref); torch.cuda.synchronize()
print('n', n, 'N', n * 6, 'of', 1 << 30, 'M', lat.M, 'build s', round(time.time() - t0, 2), 'dev MB', round(lat.device_bytes / 2**20), flush=True)
x = torch.rand((n, L), device=dev)
y = lat.filter(x); torch.cuda.synchronize()
t0 = time.time(); y = lat.filter(x, out=y); torch.cuda.synchronize()
print('filter ms', (time.time() - t0) * 1e3, 'finite', bool(torch.isfinite(y).all()), flush=True)
ye = lat.filter(x, exact=True)
rel = float(((y - ye).abs() / ye.abs().clamp_min(1e-3 * float(ye.max()))).max())
print('default vs exact max rel', rel, flush=True)
y1 = lat.filter(x[:, 1:2].contiguous())
one = float((y1[:, 0] - y[:, 1]).abs().max()) / float(y.abs().max())
print('one channel alone vs inside four: max err / max', one, flush=True)
# one more lattice on a constant image of the same size: a single cell column of the grid, the longest lists the build can see
bad = rel > 1e-4 or one > 1e-6
print('LIMIT CHECK', 'FAILED' if bad else 'ok', flush=True)
sys.exit(1 if bad else 0)

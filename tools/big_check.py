"""Robustness probe beyond the benchmark's size (tools, not a test: minutes of GPU time, tens of GB).

  python tools/big_check.py            4x the benchmark's pixel count at L = 64 (12.6 M pixels): build, filter, properties
  python tools/big_check.py wide       the same image at L = 256: 3.2 G elements = 12.9 GB per volume, i.e. element indices
                                       beyond 2^31 and byte offsets beyond 2^33 in every kernel of the step; each block of 64
                                       channels is compared with the L = 64 filter of that block (the per-channel arithmetic
                                       does not depend on the width), the exact mode with the default one, and the fused
                                       compatibility + softmax kernel with torch on a row sample from both ends of the volume
"""
import sys, time
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import torch, phl, bench

wide = len(sys.argv) > 1 and sys.argv[1] == 'wide'
H, W, L = 3072, 4096, (256 if wide else 64)
feat = bench.synthetic_features(H, W)
dev = torch.device('cuda')
t0 = time.time(); lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev)); torch.cuda.synchronize()
print('n', H * W, 'L', L, 'elements', H * W * L, 'M', lat.M, 'M/n', lat.M / (H * W), 'build s', time.time() - t0,
      'dev MB', lat.device_bytes / 2**20, lat.tile_stats(L), flush=True)
x = torch.rand((H * W, L), device=dev)
y = lat.filter(x); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): y = lat.filter(x, out=y)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print('filter ms', dt * 1e3, 'Mpl/s', H * W * L / dt / 1e6, flush=True)
print('finite', bool(torch.isfinite(y).all()), 'min', float(y.min()), flush=True)
bad = 0
if wide:
    for b in range(L // 64):
        yb = lat.filter(x[:, 64 * b:64 * (b + 1)].contiguous())
        ref = y[:, 64 * b:64 * (b + 1)]
        scale = float(ref.abs().max())
        err = float((yb - ref).abs().max()) / scale
        same = bool(torch.equal(yb, ref))
        print(f'channels {64 * b}..{64 * b + 63}: bitwise equal to the L=64 filter of the block: {same}, max err / max {err:.2e}', flush=True)
        bad += err > 1e-6
        del yb
    # strided views (row stride L, 64 channels): same answer through the staging path
    yv = lat.filter(x[:, 64:128])
    print('strided view of 64 channels vs block:', float((yv - y[:, 64:128]).abs().max()) / float(y[:, 64:128].abs().max()), flush=True)
    del yv
ye = lat.filter(x, exact=True)
rel = float(((y - ye).abs() / ye.abs().clamp_min(1e-3 * float(ye.max()))).max())
print('default vs exact max rel', rel, flush=True)
bad += rel > 1e-4
if wide:
    del ye
    mu = torch.rand((L, L), device=dev) / (L * float(y.max()))       # energies of order one: an f32 product's rounding stays below 1e-6
    e0 = torch.rand((H * W, L), device=dev)
    q = phl.compat_softmax(e0, y, mu)
    n = H * W
    for lo in (0, n // 2 - 512, n - 1024):
        rows = slice(lo, lo + 1024)
        want = torch.softmax(-(e0[rows].double() + y[rows].double() @ mu.double()), dim=1)
        err = float((q[rows].double() - want).abs().max())
        print(f'compat_softmax rows {lo}..: max abs err {err:.2e}', flush=True)
        bad += err > 1e-5
print('BIG CHECK', 'FAILED' if bad else 'ok', flush=True)
sys.exit(1 if bad else 0)

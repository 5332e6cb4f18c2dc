"""Filter step against the feature dimension d (the reference's benchmarking notebook runs d = 10: positions, colours and five
projected network features): C2 image size, L = 48 and 256; build time, M / n, stage times, staged or gather kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch, bench, phl
H, W = 1110, 1390
dev = torch.device('cuda')
base = bench.synthetic_features(H, W).reshape(-1, 5)
rng = np.random.default_rng(0)
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for d in (3, 5, 8, 10, 12):
    if d <= 5:
        feat = base[:, :d]
    else:      # extra smooth feature channels, like projected network features scaled down (benchmarking.ipynb: / 10)
        extra = np.stack([bench.synthetic_features(H, W, sigma_xy=8.0 + 3 * k).reshape(-1, 5)[:, 2 + k % 3] for k in range(d - 5)], axis=1) / 3
        feat = np.concatenate([base, extra.astype(np.float32)], axis=1)
    ref = torch.from_numpy(np.ascontiguousarray(feat)).to(dev)
    tb = t(lambda: phl.Lattice(ref), 3)
    lat = phl.Lattice(ref)
    for L in (48, 256):
        x = torch.rand((H * W, L), device=dev)
        out = torch.empty_like(x)
        ms = t(lambda: lat.filter(x, out=out))
        st = lat.tile_stats(L)
        print(f'd={d:2d} L={L:3d}: M/n {lat.M / (H * W):.3f}, build {tb:.2f} ms, filter {ms:.3f} ms = {H * W * L / ms / 1e6:.0f} G pixel-labels/s, '
              f'chunk px {st["pixels_per_chunk"]}, staged {st["staged_splat"]}/{st["staged_slice"]}, max local vertices {st["max_local_vertices"]}', flush=True)

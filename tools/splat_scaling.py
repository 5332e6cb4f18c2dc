"""k_splat_tiled time against the number of chunks in the launch (phl_splat_part on the C3 lattice with the first K
chunks): where does the small-grid penalty of the row bands come from?  python tools/splat_scaling.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd")); sys.path.insert(0, ROOT)
import torch
import bench, phl
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda")
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev))
src = bench.synthetic_values(torch, H, W, L, 0, dev)
vert = torch.empty((lat.M, L), device=dev)
partial = torch.empty((max(lat.partial_rows, 1), L), device=dev)
norows = torch.empty(0, dtype=torch.int32, device=dev)
nch = lat.tile_stats(L)["chunks"]
def t(chunks, reps=20):
    for _ in range(3): lat.splat_part(src, vert, partial, chunks, norows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lat.splat_part(src, vert, partial, chunks, norows)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for K in (128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 12288):
    ch = torch.arange(K, dtype=torch.int32, device=dev)
    us = t(ch)
    print(f"{K:6d} chunks ({K / 512:5.2f} rounds of 512 workgroup slots): {us:8.1f} us = {us / K * 512:6.1f} us per round, {K * 256 * L * 4 / us / 1e6:6.2f} TB/s of Q", flush=True)
# the same number of chunks spread over the image (every 8th chunk) instead of contiguous
ch = torch.arange(0, nch, 8, dtype=torch.int32, device=dev)
print(f"{ch.numel():6d} chunks, every 8th: {t(ch):8.1f} us")

# a 192-row band of the same image as its own lattice (what one of 8 ranks holds)
feat = bench.synthetic_features(H, W)
band = phl.Lattice(torch.from_numpy(feat[768:960].reshape(-1, 5).copy()).to(dev))
bsrc = src[768 * W:960 * W].contiguous()
bvert = torch.empty((band.M, L), device=dev)
bpart = torch.empty((max(band.partial_rows, 1), L), device=dev)
nb = band.tile_stats(L)["chunks"]
lat, src, vert, partial = band, bsrc, bvert, bpart
print(f"band lattice: {nb} chunks, M = {band.M}: {t(torch.arange(nb, dtype=torch.int32, device=dev)):8.1f} us (k_splat_tiled only)")

"""Does the relative placement of the two blur ping-pong buffers matter?  Times phl_blur (three two-axis passes) on C3
with the second buffer at controlled byte offsets from the first.  python tools/blur_align.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd")); sys.path.insert(0, ROOT)
import torch
import bench, phl
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda")
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev))
M = lat.M
rows_bytes = M * L * 4
pool = torch.empty((2 * rows_bytes + (64 << 20)) // 4, dtype=torch.float32, device=dev)
base = pool.data_ptr()
a_off = (-base) % (2 << 20)                     # first buffer on a 2 MiB boundary
def view(off_bytes):
    o = off_bytes // 4
    return pool[o:o + M * L].view(M, L)
a = view(a_off)
a.normal_()
def t(b, reps=10):
    lat.blur(a, b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lat.blur(a, b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gap0 = (rows_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)       # next 2 MiB boundary behind a
extras = (0, 1024, 16384, 0, 65536, 0, 1 << 20, 256, 0, 16384, 1024, 0)
for _ in range(3):
    t(view(a_off + gap0 + 16384))                     # warm the clocks
for extra in extras:
    b = view(a_off + gap0 + extra)
    print(f"b - a = 2 MiB-aligned gap + {extra:>9d} B: {t(b):.4f} ms", flush=True)

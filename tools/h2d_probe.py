"""Pageable host tensor -> device: phl.to_device's pinned staging pieces against plain .to(device), by size (and the way back)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import torch, phl
dev = torch.device('cuda')
def t(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for mb in (2, 7, 32, 128, 1024):
    x = torch.rand((mb << 20) // 4)
    reps = 20 if mb <= 32 else 4
    a = t(lambda: phl.to_device(x, dev), reps)
    b = t(lambda: x.to(dev), reps)
    xd = x.to(dev)
    c = t(lambda: phl.to_host(xd), reps)
    d = t(lambda: xd.cpu(), reps)
    print(f'{mb:5d} MB: to_device {a:8.3f} ms ({mb / 1024 / a * 1e3:5.1f} GB/s)  .to(device) {b:8.3f} ms ({mb / 1024 / b * 1e3:5.1f} GB/s)   '
          f'to_host {c:8.3f} ms ({mb / 1024 / c * 1e3:5.1f} GB/s)  .cpu() {d:8.3f} ms ({mb / 1024 / d * 1e3:5.1f} GB/s)', flush=True)

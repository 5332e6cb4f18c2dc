"""Pageable host tensor -> device with a plain .to(device) against 8 MB pieces, 4 ... 256 MB (no slow size band: 43-52 GB/s throughout)."""
import sys, time, torch
dev = torch.device('cuda')
def t(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for mb in (4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256):
    x = torch.rand((mb << 20) // 4)
    b = t(lambda: x.to(dev), 10)
    def chunked():
        out = torch.empty(x.shape, device=dev)
        step = (8 << 20) // 4
        for a in range(0, x.numel(), step):
            out[a:a + step].copy_(x[a:a + step], non_blocking=True)
        return out
    c = t(chunked, 10)
    print(f'{mb:4d} MB: .to {b:7.3f} ms ({mb / 1024 / b * 1e3:5.1f} GB/s)   8 MB pieces {c:7.3f} ms ({mb / 1024 / c * 1e3:5.1f} GB/s)', flush=True)

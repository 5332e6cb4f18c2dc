"""PCIe-inclusive rate of one filter call when the caller hands over HOST tensors (the reference's calling
convention: CPU in, CPU out).  Not the benchmark's `value` (that is HBM-resident), only the DESIGN.md note."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import phl

H, W, L, _ = bench.WORKLOADS["c3"]
ref = torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).cuda()
lat = phl.Lattice(ref)
src = bench.synthetic_values(torch, H, W, L, 0, torch.device("cuda")).cpu()
for pinned in (False, True):
    s = src.pin_memory() if pinned else src
    lat.filter(s)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = lat.filter(s)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"host tensors ({'pinned' if pinned else 'pageable'} input): {dt * 1e3:.0f} ms per call -> {H * W * L / dt / 1e6:.0f} Mpixel-labels/s")

"""Time phl_cost_volume at the Middlebury-sized and Tsukuba-sized problems (output-write bound: 4*h*w*L bytes)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
import numpy as np
import torch

import phl

for (h, w) in ((288, 384), (1110, 1390), (1536, 2048)):
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.random((h, w, 3), dtype=np.float32)).cuda()
    b = torch.from_numpy(rng.random((h, w, 3), dtype=np.float32)).cuda()
    L = w // 6
    out = torch.empty((h * w, L), device="cuda")
    for _ in range(3):
        phl.cost_volume(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        phl.cost_volume(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = h * w * L * 4 / 1e9
    line = f"{w}x{h}x{L}: {ms:.3f} ms, {gb:.2f} GB written -> {gb / ms * 1e3:.0f} GB/s"
    if h <= 288:
        from crf import depth
        t0 = time.time()
        depth.disparity_badness(a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
        line += f"; numpy/scipy mirror of the reference on the host: {(time.time() - t0) * 1e3:.0f} ms"
    print(line)

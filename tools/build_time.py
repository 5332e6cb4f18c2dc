"""Lattice build cost on the C3 features: wall time per build (warm scratch) for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import phl

H, W, L, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
ref = torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).cuda()
phl.Lattice(ref)
torch.cuda.synchronize()
t0 = time.time()
reps = 10
for _ in range(reps):
    lat = phl.Lattice(ref)
torch.cuda.synchronize()
print(f"{W}x{H}: {(time.time() - t0) / reps * 1e3:.2f} ms per build (warm), M = {lat.M}")

"""Build time of the reference-table lattice vs the clean one on the bench features (warm), with the library's own
timing of the host replay (PHL_DEBUG=1 prints it to stderr): python tools/reftable_time.py [workload]"""
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
for table in (False, True):
    phl.Lattice(ref, reference_table=table).close()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        t0 = time.time()
        lat = phl.Lattice(ref, reference_table=table)
        torch.cuda.synchronize()
        best = min(best, (time.time() - t0) * 1e3)
        M = lat.M
        lat.close()
    print(f"{W}x{H} reference_table={table}: {best:.2f} ms per build (warm), M = {M}")

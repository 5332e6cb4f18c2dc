"""Only the band steps of tools/band_time.py (for rocprofv3 --kernel-trace --stats): python tools/band_prof.py [world] [rank] [variant]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import bench
from loopback_dist import build_jobs

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
variant = sys.argv[3] if len(sys.argv) > 3 else "edge"
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
if variant.startswith("groups"):
    os.environ["PHL_ROWTILE_EDGE_FIRST"] = "0"
jobs, _ = build_jobs(feat, L, world, dev, want_ranks=[rank], **({"groups": int(variant[6:])} if variant.startswith("groups") else {}))
job = jobs[rank]
b = job.band
src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
out = torch.empty_like(src)
job._stub_exchange = True
for _ in range(5):
    job.filter(src, out=out)
torch.cuda.synchronize()
print("PROFILE_START", job.describe(), flush=True)
for _ in range(30):
    job.filter(src, out=out)
torch.cuda.synchronize()

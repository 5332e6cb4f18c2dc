"""Only the band steps of tools/band_time.py (for rocprofv3 --kernel-trace --stats): python tools/band_prof.py [world] [rank] [groups]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
import phl
from phl import rowtile

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
want = [r for r in (rank - 1, rank, rank + 1) if 0 <= r < world]
bands = {r: rowtile.RowBand(feat, r, world, phl.Lattice, dev) for r in want}
out = {r: b.build_outbox() for r, b in bands.items()}
b = bands[rank]
b.build_inbox({p: out[p][rank] for p in b.sides})
src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
cuts = [(g * L // groups, (g + 1) * L // groups) for g in range(groups)]
b.eng.reserve(max(c1 - c0 for c0, c1 in cuts))
res = torch.empty((b.n_local, L), device=dev)
total = sum(b.recv_rows(p) for p in b.sides)
pack = [torch.randn((total, c1 - c0), device=dev) for c0, c1 in cuts]
fake = [{p: pk[slice(*b.recv_range(p))] for p in b.sides} for pk in pack]
vert = [torch.empty((b.M, c1 - c0), device=dev) for c0, c1 in cuts]
sbuf = [torch.empty((b.send_rows(), c1 - c0), device=dev) for c0, c1 in cuts]
scr = [torch.empty((b.M, c1 - c0), device=dev) for c0, c1 in cuts]
torch.cuda.synchronize()
print("PROFILE_START", flush=True)
for _ in range(30):
    for gi, (c0, c1) in enumerate(cuts):
        b.splat_outbox(src[:, c0:c1], vert=vert[gi], sendbuf=sbuf[gi])
    for gi, (c0, c1) in enumerate(cuts):
        b.finish(vert[gi], fake[gi], out=res[:, c0:c1], packed=pack[gi], scratch=scr[gi])
torch.cuda.synchronize()

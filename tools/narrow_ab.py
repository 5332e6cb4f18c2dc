"""Chunk (LDS-staged) kernels against the gather kernels by value width, at a workload's geometry: where is the crossover?"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
import phl

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
H, W, _, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
opt = {}
if os.environ.get("TSUKUBA"):
    opt["tsukuba"] = tuple(float(x) for x in os.environ["TSUKUBA"].split(","))
feat, desc = bench.features_for(H, W, **opt)
lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, feat.shape[-1])).to(dev), reference_table=True)
print(wl, desc, "M/n", round(lat.M / (H * W), 4), flush=True)
for vd in (4, 8, 16, 32, 64, 128):
    x = torch.rand((H * W, vd), device=dev)
    out = torch.empty_like(x)
    res = {}
    for name, kw in (("chunk", {}), ("gather", {"no_tiles": True})):
        lat.reserve(vd)
        for _ in range(5):
            lat.filter(x, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lat.filter(x, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    print(f"vd {vd:4d}: chunk kernels {res['chunk']:.4f} ms, gather kernels {res['gather']:.4f} ms  (staged: {lat.tile_stats(vd)['staged_splat']}/{lat.tile_stats(vd)['staged_slice']})", flush=True)

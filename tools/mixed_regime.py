"""VERDICT r3 item 9's acceptance measurement: a half-smooth / half-iid image against the area-weighted mix of the two pure
regimes (C2 by default).  The staged-vs-gather decision is per lattice (S_multi <= 2n for the splat, S <= 3n for the slice);
the question is whether an image that mixes chunk kinds pays more than its parts."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import phl

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
H, W, L, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
smooth = bench.synthetic_features(H, W)
iid = bench.synthetic_features(H, W, iid=True)
mixed = smooth.copy()
mixed[:, W // 2:, 2:] = iid[:, W // 2:, 2:]
src = bench.synthetic_values(torch, H, W, L, 0, dev)
out = torch.empty_like(src)
res = {}
for name, feat in (("smooth", smooth), ("iid", iid), ("half smooth / half iid", mixed)):
    lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True)
    lat.reserve(L)
    for _ in range(4):
        lat.filter(src, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lat.filter(src, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    st = lat.tile_stats(L)
    tms = bench.stage_times(torch, lat, src, out, dict(exact=False, no_tiles=False), 5)
    res[name] = ms
    print(f"{name:24s}: {ms:.3f} ms  M/n {lat.M / (H * W):.3f}  multi-chunk slots / n {st['multi_chunk_slots'] / (H * W):.2f}  slots / n {st['slots'] / (H * W):.2f}  "
          f"staged {st['staged_splat']}/{st['staged_slice']}  stages {({k: round(v, 3) for k, v in tms.items()})}", flush=True)
    lat.close()
mix = 0.5 * res["smooth"] + 0.5 * res["iid"]
print(f"area-weighted mix of the pure regimes: {mix:.3f} ms; mixed image / mix = {res['half smooth / half iid'] / mix:.3f}")

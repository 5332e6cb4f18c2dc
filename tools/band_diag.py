"""Diagnostic: one band's engine built both ways (cut from the whole lattice / from its own pixels): chunk statistics,
row locality of the blur neighbours, stage times."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import bench
import phl
from phl import rowtile

world, rank = 8, 3
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
for table in ("clean", "reference"):
    if table == "clean":
        bands = {r: rowtile.RowBand(feat, r, world, phl.Lattice, dev, table="clean") for r in (rank - 1, rank, rank + 1)}
        out = {r: b.build_outbox() for r, b in bands.items()}
        b = bands[rank]
        b.build_inbox({p: out[p][rank] for p in b.sides})
        for r in (rank - 1, rank + 1):
            bands[r].build_inbox({rank: out[rank][r]})
        order = {r: bb.order_outbox() for r, bb in bands.items()}
        b.order_inbox({p: order[p][rank] for p in b.sides})
    else:
        b = rowtile.RowBand(feat, rank, world, phl.Lattice, dev, table="reference")
    eng = b.eng
    st = eng.tile_stats(L)
    nb = eng.neighbors().astype(np.int64)
    rows = eng.vertex_rows().cpu().numpy().astype(np.int64)
    ok = nb[:, :b.M_own, :] >= 0
    dist = np.abs(rows[np.clip(nb[:, :b.M_own, :], 0, None)] - rows[:b.M_own][None, :, None])
    src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
    out_t = torch.empty_like(src)
    eng.reserve(L)
    tms = bench.stage_times(torch, eng, src, out_t, dict(exact=False, no_tiles=False), 10)
    print(table, "M", eng.M, "own", b.M_own, st, "median |row - nbr row| per axis", [int(np.median(dist[a][ok[a]])) for a in range(6)],
          "stage ms", {k: round(v, 4) for k, v in tms.items()}, "blur rows", None if b.blur_rows is None else b.blur_rows[:, :, 1].tolist(), flush=True)

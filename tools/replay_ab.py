"""The analytic replay of the reference's table (default) against the simulation (PHL_REPLAY_FAST=0) on full-size
feature sets: vertex count, keys, per-pixel vertices and blur neighbours must be identical.
    python tools/replay_ab.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
import phl

cases = [("c3", dict()), ("c3", dict(sigma_xy=3.0)), ("c3", dict(tsukuba=(0.125, 0.01))), ("c2", dict()), ("c2", dict(tsukuba=(0.08, 0.03))),
         ("c5", dict()), ("c3", dict(xyd=1.0)), ("c2", dict(iid=True))]
bad = 0
for wl, opt in cases:
    H, W, L, _ = bench.WORKLOADS[wl]
    feat, desc = bench.features_for(H, W, **opt)
    ref = torch.from_numpy(feat.reshape(-1, feat.shape[-1])).cuda()
    res = {}
    for mode in ("1", "0"):
        os.environ["PHL_REPLAY_FAST"] = mode
        lat = phl.Lattice(ref, reference_table=True)
        res[mode] = (lat.M, lat.keys(), lat.replay()[0], lat.neighbors())
        lat.close()
    a, b = res["1"], res["0"]
    same = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    clean = phl.Lattice(ref).M
    print(f"{wl} {desc}: M clean {clean}, reference {a[0]}; analytic == simulation: {same}")
    bad += not same
os.environ.pop("PHL_REPLAY_FAST", None)
sys.exit(1 if bad else 0)

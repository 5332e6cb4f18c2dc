"""Experiment of VERDICT r2 item 4: the chunk splat's in-kernel reduction of shared vertices with write-through (sc1)
partial rows and one agent-scope atomic per vertex (PHL_SPLAT_FUSED=1) against the default k_splat_reduce launch.
Bitwise comparison of the vertex sums, a repeatability loop under background traffic, and interleaved timing.
    python tools/fused_check.py [workload] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
import torch

import bench
import phl

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W, L, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
for name, opt in (("sxy8", dict()), ("sxy3", dict(sigma_xy=3.0)), ("tsu_.08_.03", dict(tsukuba=(0.08, 0.03)))):
    feat, _ = bench.features_for(H, W, **opt)
    lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    src = bench.synthetic_values(torch, H, W, L, 0, dev)

    def splat(fused):
        os.environ["PHL_SPLAT_FUSED"] = "1" if fused else "0"
        return lat.splat(src)

    want = splat(False).clone()
    bad = 0
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    for it in range(reps):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        got = splat(True)
        if not torch.equal(got, want):
            bad += 1
            d = (got != want).any(1)
            print(f"  MISMATCH at {it}: {int(d.sum())} rows, max abs {float((got - want).abs().max()):.3e}")
    torch.cuda.synchronize()
    t = {}
    for fused in (False, True, False, True):
        for _ in range(5):
            splat(fused)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            splat(fused)
        e1.record()
        torch.cuda.synchronize()
        t.setdefault(fused, []).append(e0.elapsed_time(e1) / 20)
    print(f"{wl} {name}: M/n {lat.M / (H * W):.3f} mismatching launches {bad}/{reps}; splat ms default {min(t[False]):.4f} fused {min(t[True]):.4f}")
    os.environ["PHL_SPLAT_FUSED"] = "0"
    lat.close()

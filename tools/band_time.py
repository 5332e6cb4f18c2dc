"""Per-rank cost of the row-band path without a multi-GPU box: all `world` ranks of C3 are built in this process
(threads, loopback exchange), then ONE interior rank's steady-state step is timed with the exchange itself left out
(RowTileFilter._stub_exchange: every launch of the step runs, nothing travels) -- the compute-side bound on the
multi-GPU speed-up, against the whole-image step on the same box.

  python tools/band_time.py [world] [rank] [variants]

variants: comma list of  edge (default schedule), noblurrows (PHL_ROWTILE_BLUR_ROWS=0), groups1 / groups2 (channel
groups pipelined, no edge-first), staged (payloads through pinned host memory, the gloo rehearsal's schedule).
"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import bench
import phl
from loopback_dist import build_jobs

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["edge", "noblurrows", "groups1"]
H, W, L, _ = bench.WORKLOADS[os.environ.get("BAND_WORKLOAD", "c3")]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)


def gpu_ms(fn, reps=50, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, host


results = {}
for v in variants:
    env = {"noblurrows": {"PHL_ROWTILE_BLUR_ROWS": "0"}, "groups1": {"PHL_ROWTILE_EDGE_FIRST": "0"}, "groups2": {"PHL_ROWTILE_EDGE_FIRST": "0"}}.get(v, {})
    os.environ.update(env)
    kw = {"groups": 2} if v == "groups2" else ({"groups": 1} if v == "groups1" else {})
    if v.startswith("wire"):             # the default schedule with an exchange that takes <n> microseconds on the wire
        from loopback_dist import WireDist
        from phl import rowtile

        wg, rep = (int(x) for x in v[4:].split("x"))
        fake = WireDist(world, wg, rep)
        fake.local.rank = rank
        jobs = {rank: rowtile.RowTileFilter(feat, L, rank, world, dev, fake)}
    else:
        jobs, fake = build_jobs(feat, L, world, dev, want_ranks=[rank], backend="gloo" if v == "staged" else "nccl", **kw)
    for k in env:
        del os.environ[k]
    job = jobs[rank]
    b = job.band
    src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
    out = torch.empty_like(src)
    job._stub_exchange = not v.startswith("wire")
    g, h = gpu_ms(lambda: job.filter(src, out=out))
    d = job.describe()["rowtile"]
    if v.startswith("wire"):
        print(f"{v:11s}: the exchange alone ({wg} workgroups x {rep} passes over the receive buffers): {fake.wire_ms * 1e3:.0f} us", flush=True)
    print(f"{v:11s}: gpu {g:.4f} ms/step, host issue {h:.3f} ms; schedule '{d['schedule']}', M {d['M_local_plus_ghosts']} "
          f"(own {d['M_own']}), blur rows per axis {d['blur_rows_per_axis']}, edge chunks {d['edge_chunks']}", flush=True)
    results[v] = g
    del jobs, job

# single-lattice reference point on the same box
ref = torch.from_numpy(feat.reshape(-1, 5)).to(dev)
full = bench.synthetic_values(torch, H, W, L, 0, dev)
lat = phl.Lattice(ref)
lat.reserve(L)
o = torch.empty_like(full)
t1, _ = gpu_ms(lambda: lat.filter(full, out=o), reps=20, warm=10)
print(f"single lattice, whole image: {t1:.4f} ms/step")
for v, g in results.items():
    print(f"  {v:11s}: compute-side speed-up bound at {world} ranks = {t1 / g:.2f}x")

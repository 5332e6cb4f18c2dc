"""Per-rank cost of the row-band path without a multi-GPU box: build all `world` bands of C3 in this
process (loopback), then time ONE interior band's filter call (splat -> pack boundary rows -> add the
neighbours' rows -> blur -> slice) with the exchange replaced by local copies.  Reports host time
(python + launches) and GPU time per call for 1 and 2 channel groups.

  python tools/band_time.py [world] [rank] [groups,groups,...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import phl
from phl import rowtile

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
want = [r for r in (rank - 1, rank, rank + 1) if 0 <= r < world]
bands = {r: rowtile.RowBand(feat, r, world, phl.Lattice, dev) for r in want}
out = {r: b.build_outbox() for r, b in bands.items()}
b = bands[rank]
b.build_inbox({p: out[p][rank] for p in b.sides})
src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
print(f"world {world} rank {rank}: rows {b.own_rows}, S {b.S}, n_local {b.n_local}, M(+ghosts) {b.M}, "
      f"recv rows {[b.recv_rows(p) for p in b.sides]}, send rows {[int(s['send_idx'].numel()) for s in b.sides.values()]}")

print("band tiles:", b.eng.tile_stats(L))
for groups in ([int(g) for g in sys.argv[3].split(',')] if len(sys.argv) > 3 else (1, 2, 4)):
    cuts = [(g * L // groups, (g + 1) * L // groups) for g in range(groups)]
    b.eng.reserve(max(c1 - c0 for c0, c1 in cuts))
    res = torch.empty((b.n_local, L), device=dev)
    total = sum(b.recv_rows(p) for p in b.sides)
    pack = [torch.randn((total, c1 - c0), device=dev) for c0, c1 in cuts]
    fake = [{p: pk[slice(*b.recv_range(p))] for p in b.sides} for pk in pack]

    def call():
        pend = []
        for gi, (c0, c1) in enumerate(cuts):
            vert, outbox = b.splat_outbox(src[:, c0:c1])
            pend.append((vert, [v.contiguous() for v in outbox.values()]))
        for gi, (c0, c1) in enumerate(cuts):
            b.finish(pend[gi][0], fake[gi], out=res[:, c0:c1], packed=pack[gi])

    for _ in range(5):
        call()
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"groups {groups}: host issue {host:.3f} ms/call, gpu {e0.elapsed_time(e1) / reps:.3f} ms/call, wall {wall:.3f} ms/call")

# edge-first schedule (one channel group): boundary chunks, gather, [exchange], interior chunks, add, blur, slice
send = b._send_all
mask = b.eng.chunks_touching(send)
is_send = torch.zeros(b.M, dtype=torch.bool, device=dev)
is_send[send] = True
i32 = lambda m: torch.from_numpy(__import__("numpy").nonzero(m)[0].astype("int32")).to(dev)
edge, interior = i32(mask), i32(~mask)
send_rows, other_rows = torch.nonzero(is_send).flatten().to(torch.int32), torch.nonzero(~is_send).flatten().to(torch.int32)
partial = torch.empty((max(b.eng.partial_rows, 1), L), device=dev)
vert = torch.empty((b.M, L), device=dev)
scr = torch.empty_like(vert)
sbuf = torch.empty((b.send_rows(), L), device=dev)
total = sum(b.recv_rows(p) for p in b.sides)
pack1 = torch.randn((total, L), device=dev)
fake1 = {p: pack1[slice(*b.recv_range(p))] for p in b.sides}
res = torch.empty((b.n_local, L), device=dev)


def call_edge():
    b.eng.splat_part(src, vert, partial, edge, send_rows)
    b.eng.gather_rows(vert, send, out=sbuf)
    b.eng.splat_part(src, vert, partial, interior, other_rows)
    b.finish(vert, fake1, out=res, packed=pack1, scratch=scr)


for _ in range(5):
    call_edge()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(50):
    call_edge()
e1.record()
host = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print(f"edge-first ({int(edge.numel())} of {int(edge.numel() + interior.numel())} chunks feed the {int(send.numel())} boundary rows): "
      f"host issue {host:.3f} ms/call, gpu {e0.elapsed_time(e1) / 50:.3f} ms/call")

# single-lattice reference point on the same box
ref = torch.from_numpy(feat.reshape(-1, 5)).to(dev)
full = bench.synthetic_values(torch, H, W, L, 0, dev)
lat = phl.Lattice(ref)
lat.reserve(L)
print("full tiles:", lat.tile_stats(L))
o = torch.empty_like(full)
for _ in range(3):
    lat.filter(full, out=o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    lat.filter(full, out=o)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / 20 * 1e3
print(f"single lattice full image: {t1:.3f} ms/call -> compute-only speedup bound at {world} ranks = {t1:.3f}/wall")

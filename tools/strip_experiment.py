"""VERDICT r3 item 5, as a measurement before any library change: does running the splat in STRIPS of chunks, with the
partial-row reduction of a strip following right behind it on another stream, keep the partial rows on-die (Infinity
Cache: a line stays resident while the bytes touched between its two uses fit ~256 MiB) and shorten splat + reduce?

Uses only the library's existing parts API (phl_splat_part: chunk lists / row lists) from Python streams:
  strips of consecutive chunks, dealt round-robin over K chunk streams (so that consecutive strips overlap instead of
  meeting at kernel boundaries); a high-priority reduce stream completes the rows whose LAST contributing chunk lies in
  strip s as soon as strips <= s are done.  Bitwise the same vertex sums as the whole splat (checked).

  python tools/strip_experiment.py [workload] [strips,strips,...] [chunk streams]
"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import phl

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
strip_counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32, 64]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tsu = os.environ.get("TSUKUBA")
H, W, L, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
feat, desc = bench.features_for(H, W, tsukuba=tuple(float(x) for x in tsu.split(",")) if tsu else None)
lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
src = bench.synthetic_values(torch, H, W, L, 0, dev)
st = lat.tile_stats(L)
P, nch = st["pixels_per_chunk"], st["chunks"]
print(desc, "M", lat.M, st, flush=True)

# last contributing chunk of every vertex row (host, once)
t0 = time.time()
po = lat.pixel_order()
chunk_of_pixel = np.empty(lat.n, np.int32)
chunk_of_pixel[po] = np.arange(lat.n, dtype=np.int32) // P
vid, _ = lat.replay()
rows_t = lat.vertex_rows()[torch.from_numpy(vid.astype(np.int64)).to(dev)].reshape(-1)      # [n (d+1)] rows
cop_t = torch.from_numpy(chunk_of_pixel.astype(np.int64)).to(dev).repeat_interleave(vid.shape[1])
last = torch.zeros(lat.M, dtype=torch.int64, device=dev).scatter_reduce_(0, rows_t, cop_t, "amax").cpu().numpy().astype(np.int32)
first = torch.full((lat.M,), nch, dtype=torch.int64, device=dev).scatter_reduce_(0, rows_t, cop_t, "amin").cpu().numpy().astype(np.int32)
del rows_t, cop_t
print(f"host prep {time.time() - t0:.1f} s; vertices whose chunks span > 1/32 of the image: {(last - first > nch // 32).mean():.3f}", flush=True)

partial = torch.empty((max(lat.partial_rows, 1), L), device=dev)
vert = torch.empty((lat.M, L), device=dev)
none = torch.empty(0, dtype=torch.int32, device=dev)
whole = lat.splat(src).clone()


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


base = timed(lambda: lat.splat(src, out=vert))
print(f"whole splat (chunk kernel + reduce): {base:.4f} ms", flush=True)
all_ch = torch.arange(nch, dtype=torch.int32, device=dev)
all_rows = torch.arange(lat.M, dtype=torch.int32, device=dev)
t_parts = timed(lambda: (lat.splat_part(src, vert, partial, all_ch, none), lat.splat_part(src, vert, partial, none, all_rows)))
print(f"the same through the parts API, one part: {t_parts:.4f} ms", flush=True)

gw = torch.cuda.CUDAGraph()
capw = torch.cuda.Stream(device=dev)
capw.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(capw):
    lat.splat(src, out=vert)
    torch.cuda.synchronize()
    with torch.cuda.graph(gw, stream=capw):
        lat.splat(src, out=vert)
torch.cuda.synchronize()
base_g = timed(gw.replay)
print(f"whole splat replayed from a graph: {base_g:.4f} ms", flush=True)
main = torch.cuda.current_stream()
for S in strip_counts:
    per = (nch + S - 1) // S
    ch = [torch.arange(s * per, min(nch, (s + 1) * per), dtype=torch.int32, device=dev) for s in range(S)]
    strip_of = last // per
    rws = [torch.from_numpy(np.nonzero(strip_of == s)[0].astype(np.int32)).to(dev) for s in range(S)]
    xs = [torch.cuda.Stream(device=dev) for _ in range(K)]
    red = torch.cuda.Stream(device=dev, priority=-1)
    evx = [torch.cuda.Event() for _ in range(K)]
    ev0, evr = torch.cuda.Event(), torch.cuda.Event()

    def run():
        ev0.record(main)
        for x in xs:
            x.wait_event(ev0)
        red.wait_event(ev0)
        for s in range(S):
            x = xs[s % K]
            with torch.cuda.stream(x):
                lat.splat_part(src, vert, partial, ch[s], none)
                evx[s % K].record(x)
            with torch.cuda.stream(red):
                for k in range(min(K, s + 1)):
                    red.wait_event(evx[(s - k) % K])
                lat.splat_part(src, vert, partial, none, rws[s])
        evr.record(red)
        main.wait_event(evr)

    vert.fill_(float("nan"))
    run()
    torch.cuda.synchronize()
    ok = bool(torch.equal(vert, whole))
    t = timed(run)
    # the same launches replayed from one HIP graph: no host time between them (2 S launches from Python are host-bound)
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(main)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        main_saved = main
        main = cap
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            run()
        main = main_saved
    torch.cuda.synchronize()
    vert.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    ok_g = bool(torch.equal(vert, whole))
    tg = timed(g.replay)
    print(f"strips {S:3d} x {per} chunks, {K} chunk streams: eager {t:.4f} ms, graph replay {tg:.4f} ms  (whole {base:.4f} eager, "
          f"{base_g:.4f} graph; bitwise equal {ok} / {ok_g})", flush=True)

"""Debug: per-workgroup timeline of k_splat_tiled (PHL_TIMELINE dump; 100 MHz stamps).
  python tools/splat_timeline.py [rows] -> prints round structure, phase durations, per-CU utilisation."""
import os
import sys

analyze_only = len(sys.argv) > 2 and sys.argv[1] == "--analyze"      # python tools/splat_timeline.py --analyze dump.bin
path = sys.argv[2] if analyze_only else "/tmp/phl_timeline.bin"
os.environ["PHL_TIMELINE"] = path
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-estimation_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
import phl

rows = 0 if analyze_only else (int(sys.argv[1]) if len(sys.argv) > 1 else 192)
H, W, L, _ = bench.WORKLOADS["c3"]
if not analyze_only:
    dev = torch.device("cuda", 0)
    feat = bench.synthetic_features(H, W)
    r0 = (H - rows) // 2
    lat = phl.Lattice(torch.from_numpy(np.ascontiguousarray(feat[r0:r0 + rows].reshape(-1, 5))).to(dev))
    src = bench.synthetic_values(torch, rows, W, L, r0, dev)
    for _ in range(6):
        lat.splat(src)
    torch.cuda.synchronize()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 0] > 0]
st = t[:, 0].astype(np.int64)
t0 = st.min()
us = lambda x: (x.astype(np.int64) - t0) / 100.0
start, s0in, s0done, s1in, s1done, end = (us(t[:, k]) for k in range(6))
hw = t[:, 7]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
cuid = xcc * 64 + se * 16 + sh * 8 + cu
print(f"rows {rows}: {len(t)} workgroups, kernel span {end.max():.1f} us, distinct CUs {len(np.unique(cuid))}, XCCs {np.unique(xcc)}")
dur = end - start
print(f"workgroup duration: mean {dur.mean():.1f}  p10 {np.percentile(dur, 10):.1f}  p50 {np.percentile(dur, 50):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f}")
print(f"  prologue+stage slab0 {np.mean(s0in - start):.1f} | sum slab0 {np.mean(s0done - s0in):.1f} | store+wait slab1 {np.mean(s1in - s0done):.1f} | sum slab1 {np.mean(s1done - s1in):.1f}")
order = np.argsort(start)
for name, sel in (("first 512 started", order[:512]), ("last 512 started", order[-512:]), ("middle", order[len(order) // 2 - 256:len(order) // 2 + 256])):
    d = dur[sel]
    print(f"  {name}: start {start[sel].min():.1f}..{start[sel].max():.1f}  duration mean {d.mean():.1f}  "
          f"stage0 {np.mean((s0in - start)[sel]):.1f} sum0 {np.mean((s0done - s0in)[sel]):.1f} stage1 {np.mean((s1in - s0done)[sel]):.1f} sum1 {np.mean((s1done - s1in)[sel]):.1f}")
# per-CU busy time (two workgroups can overlap: count union length and sum)
tot = end.max()
busy = []
per_cu = []
for c in np.unique(cuid):
    m = cuid == c
    per_cu.append(m.sum())
    busy.append(dur[m].sum() / (2 * tot))
print(f"workgroups per CU: min {min(per_cu)} max {max(per_cu)}; slot utilisation (sum of durations / 2 slots / span): mean {np.mean(busy):.2f} min {np.min(busy):.2f}")
for x in np.unique(xcc):
    m = xcc == x
    print(f"  XCC {x}: {m.sum()} workgroups, last end {end[m].max():.1f} us, mean duration {dur[m].mean():.1f}")
hist, edges = np.histogram(end, bins=10, range=(0, tot))
print("workgroup completions per tenth of the span:", hist.tolist())

"""One rank's band step with a stand-in exchange (WireDist) for `rocprofv3 --kernel-trace`: python tools/band_trace.py W R
Afterwards: python tools/band_trace.py --analyze <kernel_trace.csv> prints the last steps' kernels with start / end relative
to the step's first kernel and the queue each ran on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    import csv

    rows = list(csv.DictReader(open(sys.argv[2])))
    ks = []
    for r in rows:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    ks.sort()
    # steps start at a k_splat_tiled whose predecessor (in time) is a k_slice_tiled
    starts = [i for i, k in enumerate(ks) if k[2].startswith("k_splat_tiled") and i > 0 and ks[i - 1][2].startswith("k_slice_tiled")]
    for si in starts[-3:-1]:
        t0 = ks[si][0]
        print("---- step")
        for k in ks[si:si + 14]:
            print(f"  {k[2][:34]:34s} queue {k[3]:>3s} stream {k[4]:>3s}  start {(k[0] - t0) / 1e3:8.1f} us  end {(k[1] - t0) / 1e3:8.1f} us")
            if k[2].startswith("k_slice_tiled"):
                break
    sys.exit(0)

import torch

import bench
from loopback_dist import WireDist
from phl import rowtile

wg, rep = int(sys.argv[1]), int(sys.argv[2])
world, rank = 8, 3
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
fake = WireDist(world, wg, rep)
fake.local.rank = rank
job = rowtile.RowTileFilter(feat, L, rank, world, dev, fake)
b = job.band
src = bench.synthetic_values(torch, b.own_rows, W, L, b.row0, dev)
out = torch.empty_like(src)
for _ in range(30):
    job.filter(src, out=out)
torch.cuda.synchronize()
print(job.describe()["rowtile"]["schedule"])

"""Per-step times of the lattice filter (C3) over 60 back-to-back steps from an idle GPU: does the clock ramp that
tools/compat_seq.py shows for the MFMA kernel matter for the HBM-bound filter step?  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, phl
H, W, L, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
feat = bench.synthetic_features(H, W)
lat = phl.Lattice(torch.from_numpy(np.ascontiguousarray(feat.reshape(-1, 5))).to(dev))
src = bench.synthetic_values(torch, H, W, L, 0, dev)
out = torch.empty_like(src)
lat.filter(src, out=out); torch.cuda.synchronize()
time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
ev[0].record()
for k in range(60):
    lat.filter(src, out=out); ev[k + 1].record()
torch.cuda.synchronize()
ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(60)]
print("per-step ms:", " ".join("%.2f" % t for t in ts))
print("mean of steps 1-20 after 3 warm-up steps: %.3f   steady (last 20): %.3f" % (np.mean(ts[3:23]), np.mean(ts[40:])))

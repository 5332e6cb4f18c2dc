"""Debug: phase timeline of k_compat_softmax's persistent workgroups (needs a library built with
`make -C depth-estimation_amd/csrc EXTRA=-DPHL_COMPAT_TIMELINE`; 100 MHz stamps of each workgroup's first 8 tiles).
  python tools/compat_timeline.py  -> durations of the MFMA and epilogue halves per wave group."""
import os, sys
path = "/tmp/phl_compat_timeline.bin"
os.environ["PHL_COMPAT_TIMELINE"] = path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import phl
n, L = 1536 * 2048, 256
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
E0 = torch.rand((n, L), device=dev, generator=g) * 10
X = torch.rand((n, L), device=dev, generator=g)
Mu = torch.rand((L, L), device=dev, generator=g) * 3
out = torch.empty_like(E0)
for _ in range(4):                                   # each launch dumps the one before it
    phl.compat_softmax(E0, X, Mu, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); phl.compat_softmax(E0, X, Mu, out=out); b.record(); torch.cuda.synchronize()
print("PHL_CS_EXP", os.environ.get("PHL_CS_EXP"), "kernel ms", a.elapsed_time(b))
t = np.fromfile(path, dtype=np.uint64)[:256 * 2 * 40].reshape(256, 2, 40).astype(np.int64)
arr = t[:, :, 16:32]
pre = t[:, :, 32:40]
t = t[:, :, :16]
t0 = t[t > 0].min()
us = (t - t0) / 100.0                                 # [workgroup, group, stamp]: stamp 2k = MFMA half of the k-th tile, 2k+1 = its epilogue half
mf = us[:, :, 1::2] - us[:, :, 0::2]
ep = us[:, :, 2::2] - us[:, :, 1:-1:2]
for k in range(7):
    print(f"tile {k}: MFMA half {mf[:, :, k].mean():6.1f} us (min {mf[:, :, k].min():5.1f} max {mf[:, :, k].max():5.1f})   epilogue half {ep[:, :, k].mean():6.1f} (min {ep[:, :, k].min():5.1f} max {ep[:, :, k].max():5.1f})")
print("group 0 of workgroup 0:", " ".join("%.0f" % v for v in us[0, 0]))
print("group 1 of workgroup 0:", " ".join("%.0f" % v for v in us[0, 1]))
# iteration 2 of workgroup 0: arrival at each slot's barrier (8 MFMA slots, then 8 epilogue slots), per group
a = (arr - t0) / 100.0
for wg in (0, 100):
    g0, g1 = a[wg, 0], a[wg, 1]
    print(f"workgroup {wg}: group 0 arrivals", " ".join("%.1f" % v for v in g0))
    print(f"workgroup {wg}: group 1 arrivals", " ".join("%.1f" % v for v in g1))
# group 0's epilogue slots (8..15) coincide with group 1's MFMA slots (0..7) of the same iteration
d = a[:, 1, 0:8] - a[:, 0, 8:16]
print("arrival(MFMA wave) - arrival(epilogue wave) per slot, mean over workgroups:", " ".join("%.2f" % v for v in d.mean(axis=0)))
print("slot length (barrier to barrier), group 1 MFMA half:", " ".join("%.2f" % v for v in np.diff(a[:, 1, 0:8], axis=1).mean(axis=0)))
print("drain stall per slot (MFMA wave, us), mean over workgroups/groups:", " ".join("%.2f" % v for v in ((arr[:, :, 0:8] - pre) / 100.0).mean(axis=(0, 1))))

"""Debug: phase timeline of k_compat_softmax's two wave groups.  Needs a library built with
`touch depth-estimation_amd/csrc/phl_meanfield.hip; make -C depth-estimation_amd/csrc EXTRA="-DPHL_COMPAT_TIMELINE"`
(100 MHz stamps at the start of each half of a group's first 8 tiles); add -DPHL_CS_FINE for shader-clock stamps of
every barrier arrival in the third iteration (each stamp costs ~800 cycles: read differences, not totals).
  python tools/compat_timeline.py  -> duration of the MFMA and epilogue halves; who arrives last at each slot's barrier."""
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
t = np.fromfile(path, dtype=np.uint64)[:256 * 2 * 64].reshape(256, 2, 64).astype(np.int64)
arr = t[:, :, 16:32]                                  # shader-clock arrival at each slot's barrier in the third iteration: 8 MFMA slots, 8 epilogue slots
t = t[:, :, :16]
t0 = t[t > 0].min()
us = (t - t0) / 100.0                                 # [workgroup, group, stamp]: stamp 2k = MFMA half of the k-th tile, 2k+1 = its epilogue half
mf = us[:, :, 1::2] - us[:, :, 0::2]
ep = us[:, :, 2::2] - us[:, :, 1:-1:2]
for k in range(7):
    print(f"tile {k}: MFMA half {mf[:, :, k].mean():6.1f} us (min {mf[:, :, k].min():5.1f} max {mf[:, :, k].max():5.1f})   epilogue half {ep[:, :, k].mean():6.1f} (min {ep[:, :, k].min():5.1f} max {ep[:, :, k].max():5.1f})")
# group 0's epilogue slots (8..15) are group 1's MFMA slots (0..7) of the same iteration
d = (arr[:, 1, 0:8] - arr[:, 0, 8:16]).astype(np.float64)
print("arrival(MFMA wave) - arrival(epilogue wave), shader cycles per slot:", " ".join("%6.0f" % v for v in d.mean(axis=0)))
print("slot length (MFMA wave arrival to arrival):", " ".join("%6.0f" % v for v in np.diff(arr[:, 1, 0:8].astype(np.float64), axis=1).mean(axis=0)))

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
t = np.fromfile(path, dtype=np.uint64)[:256 * 2 * 64].reshape(256, 2, 64).astype(np.int64)
fine = t[:, :, 16:48].reshape(256, 2, 8, 4)          # shader-clock stamps of the third tile's 8 chunks: entry, prefetch issued, MFMAs done, drained
t = t[:, :, :16]
t0 = t[t > 0].min()
us = (t - t0) / 100.0                                 # [workgroup, group, stamp]: stamp 2k = MFMA half of the k-th tile, 2k+1 = its epilogue half
mf = us[:, :, 1::2] - us[:, :, 0::2]
ep = us[:, :, 2::2] - us[:, :, 1:-1:2]
for k in range(7):
    print(f"tile {k}: MFMA half {mf[:, :, k].mean():6.1f} us (min {mf[:, :, k].min():5.1f} max {mf[:, :, k].max():5.1f})   epilogue half {ep[:, :, k].mean():6.1f} (min {ep[:, :, k].min():5.1f} max {ep[:, :, k].max():5.1f})")
d = np.diff(fine, axis=3).astype(np.float64)          # per chunk: issue, MFMA stream, drain (shader cycles)
nxt = (fine[:, :, 1:, 0] - fine[:, :, :-1, 3]).astype(np.float64)   # drained -> next chunk's entry (the barrier)
print("per chunk, shader cycles (mean over workgroups and groups):")
print("  prefetch issue :", " ".join("%6.0f" % v for v in d[:, :, :, 0].mean(axis=(0, 1))))
print("  MFMA stream    :", " ".join("%6.0f" % v for v in d[:, :, :, 1].mean(axis=(0, 1))), "  (128 MFMAs = 8192)")
print("  drain          :", " ".join("%6.0f" % v for v in d[:, :, :, 2].mean(axis=(0, 1))))
print("  barrier        :", " ".join("%6.0f" % v for v in nxt.mean(axis=(0, 1))))

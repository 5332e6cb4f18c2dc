"""Upper bound on what fusing the slice into the compatibility kernel could give, measured before
writing it: phl_compat_softmax with its X operand served from L2 (row stride 0: every tile reads the same row, so X costs no
HBM traffic at all) against the real call.  The fused kernel would ADD the slice's gather work (d+1 LDS row reads and FMAs
per pixel and label) to the compatibility kernel's non-matrix half; this probe removes X's HBM read without adding anything.
If the kernel does not get faster here, it cannot get faster by fusion: the saving is then only the slice launch itself
minus whatever the gathers cost inside."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)
import phl

n, L = 1536 * 2048, 256
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
E0 = torch.rand((n, L), device=dev, generator=g) * 10
X = torch.rand((n, L), device=dev, generator=g)
labels = torch.arange(L, dtype=torch.float32, device=dev)
Mu = torch.sqrt(9.0 + (labels[:, None] - labels[None, :]) ** 2) - 3.0
out = torch.empty_like(E0)


def t(f, reps=10):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


real = t(lambda: phl.compat_softmax(E0, X, Mu, out=out))
X0 = X[:1].expand(n, L)                      # row stride 0: one 1 KB row, L2-resident
assert X0.stride(0) == 0 and X0.stride(1) == 1
nox = t(lambda: phl.compat_softmax(E0, X0, Mu, out=out))
E00 = E0[:1].expand(n, L)
noxe = t(lambda: phl.compat_softmax(E00, X0, Mu, out=out))
print(f"k_compat_softmax, C3: X from HBM {real:.3f} ms; X from L2 (stride 0) {nox:.3f} ms; X and E0 from L2 {noxe:.3f} ms")
print(f"=> removing X's 3.2 GB HBM read from the kernel is worth {real - nox:.3f} ms; the slice launch it would replace takes ~0.75 ms "
      f"and would bring 6 LDS row reads + 6 FMAs per pixel and label into the kernel's non-matrix half")

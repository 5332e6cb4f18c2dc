"""Accuracy and time of the two arithmetic forms of phl.compat_softmax (f32 matrix cores / bf16 matrix cores on operands
split three ways) against float64, on a sample of rows.  Run on the GPU box under `timeout`."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import phl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 1024 + 77
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for L in (256, 252, 228, 224, 200, 192, 160, 132):
    E0 = torch.rand((n, L), device=dev, generator=g) * 10
    X = torch.rand((n, L), device=dev, generator=g) * 4 - 1
    Mu = (torch.rand((L, L), device=dev, generator=g) - 0.3) * 0.2
    for logits in (False, True):
        want = -(E0.double() + X.double() @ Mu.double())
        if not logits:
            want = torch.softmax(want, dim=1)
        res = {}
        for arith in ('f32', 'split'):
            out = phl.compat_softmax(E0, X, Mu, logits=logits, arith=arith)
            torch.cuda.synchronize()
            res[arith] = float((out.double() - want).abs().max())
        print(f'L={L} logits={logits}: max abs err vs float64  f32 {res["f32"]:.3e}   split {res["split"]:.3e}', flush=True)
        bad += not (res['split'] <= 4 * res['f32'] + 1e-12)
print('COMPAT CHECK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)

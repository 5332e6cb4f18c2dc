"""Label counts as the reference's own formula gives them (crf/depth.py:40: max_disp = w // 6 -- 231 at Middlebury's 1390 columns,
64 at Tsukuba's 384): one mean-field iteration and its parts at L = 231 / 232 / 256 on the C2 image size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import torch, bench, phl
from crf.crf_module import charbonneir, compatibility_matrix, mean_field_infer
from crf.gaussian_matrix import LatticeGaussian
H, W = 1110, 1390
dev = torch.device('cuda')
ref = torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev)
Wop = LatticeGaussian(ref)
def t(f, reps=5):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for L in (int(a) for a in (sys.argv[1:] or (231, 232, 256))):
    g = torch.Generator(device=dev).manual_seed(L)
    E0 = torch.rand((H * W, L), device=dev, generator=g) * 10
    Mu = compatibility_matrix(lambda a, b: charbonneir(a, b, 3.0), torch.arange(L, dtype=torch.float32, device=dev))
    Q = torch.softmax(-E0, dim=1)
    X = Wop @ Q
    print(f'L={L}: W@Q {t(lambda: Wop @ Q):.3f} ms, compat_softmax {t(lambda: phl.compat_softmax(E0, X, Mu)):.3f} ms, '
          f'mean_field_infer(5 iterations) {t(lambda: mean_field_infer(E0, Wop, Mu, 5), 3):.2f} ms', flush=True)

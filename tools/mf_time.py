import sys, time, torch
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import bench, phl
from crf.crf_module import charbonneir, compatibility_matrix, mean_field_infer
from crf.gaussian_matrix import LatticeGaussian
H, W, L = 1536, 2048, 256
dev = torch.device('cuda')
feat = bench.synthetic_features(H, W)
ref = torch.from_numpy(feat.reshape(-1, 5)).to(dev)
E0 = torch.rand((H * W, L), device=dev) * 10
labels = torch.arange(L, dtype=torch.float32, device=dev)
Mu = compatibility_matrix(lambda a, b: charbonneir(a, b, 3), labels)
Wop = LatticeGaussian(ref)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
Q = torch.softmax(-E0, 1)
print('softmax ms', t(lambda: torch.softmax(-E0, 1)))
print('filter-U ms', t(lambda: Wop @ Q))
X = Wop @ Q
print('matmul fp32 ms', t(lambda: X @ Mu))
print('E0 + ... ms', t(lambda: E0 + X))
print('mean_field_infer 5 iters ms', t(lambda: mean_field_infer(E0, Wop, Mu, 5), 2))

E = torch.empty_like(E0)
print('addmm ms', t(lambda: torch.addmm(E0, X, Mu, out=E)))
print('mm ms', t(lambda: torch.mm(X, Mu)))

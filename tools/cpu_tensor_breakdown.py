"""Where the notebook's CPU-tensor call (mean_field_infer on CPU tensors at C1: 384x288x16, 5 iterations) spends its time on
this box: pageable -> device, the device loop, device -> pinned host; and the plain alternatives (.to / .cpu)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import torch, bench, phl
import crf.crf_module as cm
from crf.gaussian_matrix import LatticeGaussian
H, W, L, _ = bench.WORKLOADS['c1']
dev = torch.device('cuda')
ref = torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5))
E0 = torch.rand((H * W, L)) * 10
Mu = cm.compatibility_matrix(lambda a, b: cm.charbonneir(a, b, 3.0), torch.arange(L, dtype=torch.float32))
Wc = LatticeGaussian(ref)
def t(f, reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f'whole call (CPU in, CPU out): {t(lambda: cm.mean_field_infer(E0, Wc, Mu, 5)):.3f} ms')
print(f'  phl.to_device(E0) [{E0.numel() * 4 / 1e6:.1f} MB pageable]: {t(lambda: phl.to_device(E0, dev)):.3f} ms   E0.to(device): {t(lambda: E0.to(dev)):.3f} ms')
E0d = E0.to(dev); Wd = LatticeGaussian(ref.to(dev)); Mud = Mu.to(dev)
print(f'  device loop: {t(lambda: cm.mean_field_infer(E0d, Wd, Mud, 5)):.3f} ms')
Q = cm.mean_field_infer(E0d, Wd, Mud, 5)
print(f'  phl.to_host(Q): {t(lambda: phl.to_host(Q)):.3f} ms   Q.cpu(): {t(lambda: Q.cpu()):.3f} ms')
print(f'  lattice lookup in the cache for a CPU ref: {t(lambda: phl.lattice_for(ref)):.3f} ms')

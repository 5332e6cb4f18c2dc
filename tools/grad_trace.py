"""The fused backward (phl_filter_grad) at C3 under `rocprofv3 --kernel-trace --stats`: per-kernel time of one call.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/grad_trace -- python3 tools/grad_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import torch, bench, phl
H, W, L = 1536, 2048, 256
dev = torch.device('cuda')
ref = torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev)
lat = phl.Lattice(ref)
g = torch.Generator(device=dev).manual_seed(5)
src = torch.rand((H * W, L), device=dev, generator=g)
gout = torch.randn((H * W, L), device=dev, generator=g)
for _ in range(2):
    lat.filter_grad(src, gout, ref)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    lat.filter_grad(src, gout, ref)
b.record(); torch.cuda.synchronize()
print('filter_grad ms', a.elapsed_time(b) / 5)

"""Timing of the fused compatibility + softmax kernel (phl_compat_softmax) against rocBLAS mm + fused softmax,
on C3-size operands (3,145,728 x 256).  Run on the GPU box; prints steady-state HIP-event times (10 launches after 10
warm-up launches; tools/compat_seq.py shows the ramp: an isolated launch is ~15 % slower than the tenth in a row)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import phl
n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1536 * 2048), (int(sys.argv[2]) if len(sys.argv) > 2 else 256)
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
E0 = torch.rand((n, L), device=dev, generator=g) * 10
X = torch.rand((n, L), device=dev, generator=g)
Mu = torch.rand((L, L), device=dev, generator=g) * 3
out = torch.empty_like(E0)
def t(f, reps=10):
    for _ in range(10): f()          # steady state: the clocks take ~6 back-to-back launches (25 ms) to come up from idle
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: phl.compat_softmax(E0, X, Mu, out=out, arith='f32'))
print(f"fused compat_softmax, f32 matrix cores      n={n} L={L}: {ms:.3f} ms  = {2 * n * L * L / ms / 1e9:.1f} TFLOP/s f32, {3 * n * L * 4 / ms / 1e6:.0f} GB/s of compulsory traffic")
ms_l = t(lambda: phl.compat_softmax(E0, X, Mu, out=out, logits=True, arith='f32'))
print(f"fused, logits epilogue   : {ms_l:.3f} ms")
if 128 < L <= 256:
    ms_s = t(lambda: phl.compat_softmax(E0, X, Mu, out=out, arith='split'))
    print(f"fused compat_softmax, bf16 matrix cores on three-way split operands: {ms_s:.3f} ms  = {3 * n * L * 4 / ms_s / 1e6:.0f} GB/s of compulsory traffic, "
          f"{6 * 2 * n * L * L / ms_s / 1e9:.0f} TFLOP/s bf16")
    ms_sl = t(lambda: phl.compat_softmax(E0, X, Mu, out=out, logits=True, arith='split'))
    print(f"split, logits epilogue   : {ms_sl:.3f} ms")
G = torch.empty_like(E0)
ms_mm = t(lambda: torch.mm(X, Mu, out=G))
ms_sm = t(lambda: phl.softmax_neg_add(E0, G, out=out))
print(f"rocBLAS mm {ms_mm:.3f} ms ({2 * n * L * L / ms_mm / 1e9:.1f} TF) + fused add/softmax {ms_sm:.3f} ms = {ms_mm + ms_sm:.3f} ms")
Mp = torch.ones((L, L), device=dev) - torch.eye(L, device=dev)          # the reference's potts layer (crf_module.py:55-64)
ms_p = t(lambda: phl.compat_softmax(E0, X, Mp, out=out))
ms_pd = t(lambda: phl.compat_softmax(E0, X, Mp, out=out, structure=False, arith='f32'))
print(f"Potts compatibility (1 - I): streaming pass {ms_p:.3f} ms = {3 * n * L * 4 / ms_p / 1e6:.0f} GB/s of E0 + X + Q; as a dense product {ms_pd:.3f} ms")

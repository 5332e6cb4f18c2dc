"""Race hunt for the LDS rings and counted waits of k_compat_softmax and k_compat_split: 150 launches per shape beside background memory traffic,
every result compared bit for bit with the first (run on the GPU box: python tools/compat_stress.py)."""
import os, sys, torch
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import phl
g = torch.Generator(device='cuda').manual_seed(1)
bad = 0
for (n, L, arith) in ((1536 * 2048, 256, 'split'), (128 * 4001 + 3, 252, 'split'), (1536 * 2048, 256, 'f32'), (128 * 4001 + 3, 224, 'f32'), (128 * 9000, 96, 'f32')):
    E0 = torch.rand((n, L), device='cuda', generator=g) * 20
    X = torch.rand((n, L), device='cuda', generator=g)
    Mu = torch.rand((L, L), device='cuda', generator=g) * 2
    ref = phl.compat_softmax(E0, X, Mu, arith=arith).clone()
    want = torch.softmax(-(E0 + X @ Mu), dim=1)
    print(n, L, arith, 'err vs torch', float((ref - want).abs().max()), flush=True)
    out = torch.empty_like(ref)
    # background traffic on another stream to vary memory latency
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, device='cuda', dtype=torch.uint8)
    for it in range(150):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        phl.compat_softmax(E0, X, Mu, out=out, arith=arith)
        if not torch.equal(out, ref):
            bad += 1
            print('MISMATCH at', it, float((out - ref).abs().max()))
    torch.cuda.synchronize()
print('mismatches:', bad)
sys.exit(1 if bad else 0)

"""Seeded fuzz of Lattice(reference_table=True) -- the analytic replay of the reference's hash-table doublings, its speculative
device queries included -- against the oracle's faithful-table mode (== the reference engine bit for bit): vertex count, keys,
per-pixel vertices and weights, blur neighbours, exact-mode filter output.  python tools/fuzz_reftable.py SEED0 COUNT.
With PHL_DEBUG=1 the library says which replay (analytic / simulation) each lattice took."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'depth-estimation_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch, phl
from oracle import phl_oracle as po

seed0, count = int(sys.argv[1]), int(sys.argv[2])
bad = done = 0
t0 = time.time()
for seed in range(seed0, seed0 + count):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([6000, 9000, 20000, 47000, 90000, 150000]))
    d = int(rng.choice([2, 3, 4, 5, 6]))
    scale = float(rng.choice([3.0, 6.0, 12.0, 30.0]))
    ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    if rng.integers(0, 3) == 0:                       # image-like: smooth in the first two dimensions
        side = int(np.sqrt(n))
        n = side * side
        yy, xx = np.mgrid[:side, :side].astype(np.float32)
        ref = np.concatenate([np.stack([yy, xx], -1).reshape(n, 2) / np.float32(rng.choice([1.5, 3.0])),
                              rng.random((n, max(1, d - 2)), dtype=np.float32) * np.float32(scale / 3)], axis=1).astype(np.float32)
        d = ref.shape[1]
    src = rng.standard_normal((n, 4)).astype(np.float32)
    O = po.Oracle(ref, faithful_table=True)
    if O.status == 1 or O.M < 16383:
        continue
    want = O.filter(src)
    L = phl.Lattice(torch.from_numpy(ref).cuda(), reference_table=True)
    done += 1
    ok = L.M == O.M and np.array_equal(L.keys(), O.keys())
    if ok:
        vid, w = L.replay()
        ovid, ow = O.replay()
        ok = np.array_equal(vid, ovid) and np.array_equal(w.view(np.uint32), ow.view(np.uint32)) and np.array_equal(L.neighbors(), O.neighbors())
        ok = ok and np.array_equal(L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy().view(np.uint32), want.view(np.uint32))
    if not ok:
        bad += 1
        print('MISMATCH', dict(seed=seed, n=n, d=d, scale=scale, M=int(O.M), M_gpu=int(L.M)), flush=True)
    if done % 10 == 0:
        print(f'... {done} lattices, {bad} mismatches, {time.time() - t0:.0f} s', flush=True)
print(f'fuzz: {done} reference-table lattices over seeds {seed0}..{seed0 + count - 1}, {bad} mismatches')
sys.exit(1 if bad else 0)

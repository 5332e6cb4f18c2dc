"""GPU parity of the HIP lattice (through the C ABI) against the CPU oracle and the committed
golden vectors generated from the reference itself.

Bar (task statement): bit-exact for integer / index work (keys, vertex numbering, replay
offsets, neighbour ids, contribution lists); floating point within 1e-4 relative -- and in
fact bit-exact too, because every stage is written to round like the reference's scalar
loops (see csrc/phl_filter.hip header).  Both bars are asserted: exactness as the primary
check, RTOL as the documented tolerance.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star tolerance: "within 1e-4 relative per pixel"


def rel_err(a, b):
    scale = np.maximum(np.abs(b), 1e-3 * max(np.abs(b).max(), 1e-30))
    return float((np.abs(a - b) / scale).max()) if a.size else 0.0


def scaled_err(a, b):
    """max abs error relative to the largest magnitude: the right yardstick for a changed
    summation order on signed data (per-element relative error is unbounded under cancellation)."""
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


@pytest.fixture(scope="module")
def phl():
    import phl as _phl

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _phl.load_library()
    return _phl


def _rand_case(n, d, vd, scale, seed):
    rng = np.random.default_rng(seed)
    ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    src = rng.standard_normal((n, vd)).astype(np.float32)
    return ref, src


CASES = [(64, 2, 1, 3.0), (500, 1, 2, 10.0), (2000, 3, 3, 5.0), (2000, 5, 16, 4.0), (3000, 8, 5, 2.0),
         (4096, 5, 64, 2.0), (1000, 2, 7, 300.0), (5000, 5, 256, 3.0), (20000, 5, 4, 8.0), (3000, 16, 8, 1.0),
         (7000, 4, 20, 2.5), (60000, 5, 12, 6.0), (1, 5, 4, 1.0), (3, 2, 3, 0.0),
         (2500, 7, 3, 2.0), (1500, 11, 2, 1.5), (1200, 13, 2, 1.0), (2000, 6, 4, 3.0)]   # (record packing at every width)


@pytest.mark.parametrize("n,d,vd,scale", CASES)
def test_build_matches_oracle_exactly(phl, n, d, vd, scale):
    from oracle import phl_oracle as po

    ref, _ = _rand_case(n, d, vd, scale, 100 + n)
    O = po.Oracle(ref)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    assert L.M == O.M
    assert np.array_equal(L.keys(), O.keys()), "vertex keys / first-touch numbering"
    vid, w = L.replay()
    ovid, ow = O.replay()
    assert np.array_equal(vid, ovid), "replay vertex ids"
    assert np.array_equal(w.view(np.uint32), ow.view(np.uint32)), "barycentric weights (bitwise)"
    assert np.array_equal(L.neighbors(), O.neighbors()), "blur neighbour table"
    # contribution lists: per vertex, ascending pixel, weights = transposed replay
    ptr, pix, cw = L.splat_lists()
    assert ptr[0] == 0 and ptr[-1] == n * (d + 1)
    order = np.lexsort((np.repeat(np.arange(n), d + 1), ovid.ravel()))
    assert np.array_equal(pix, np.repeat(np.arange(n), d + 1)[order].astype(np.int32))
    assert np.array_equal(cw.view(np.uint32), ow.ravel()[order].view(np.uint32))
    assert np.array_equal(np.diff(ptr), np.bincount(ovid.ravel(), minlength=O.M))


@pytest.mark.parametrize("n,d,vd,scale", CASES)
def test_filter_stages_match_oracle(phl, n, d, vd, scale):
    from oracle import phl_oracle as po

    ref, src = _rand_case(n, d, vd, scale, 100 + n)
    O = po.Oracle(ref)
    out_o, splat_o, blur_o = O.filter(src, stages=True)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    s = torch.from_numpy(src).cuda()
    # pixel-ordered gather splat: the reference's summation order, bit for bit
    ft = lambda v: L.to_first_touch(v).cpu().numpy()     # vertex buffers are in internal row order; the oracle's in first-touch order
    vs = L.splat(s, exact=True)
    assert np.array_equal(ft(vs).view(np.uint32), splat_o.view(np.uint32)), "splat bitwise"
    # default LDS-staged chunk splat: per-chunk partial sums -> fp32 rounding only
    vt = L.splat(s)
    assert scaled_err(ft(vt), splat_o) <= 1e-5
    assert rel_err(ft(L.splat(s, no_tiles=True)), splat_o) == 0.0
    vb = L.blur(vs)
    assert np.array_equal(ft(vb).view(np.uint32), blur_o.view(np.uint32)), "blur bitwise"
    for no_tiles in (False, True):                  # LDS-staged and plain gather slice: both bit-exact
        out = L.slice(vb, exact=True, no_tiles=no_tiles).cpu().numpy()
        assert np.array_equal(out.view(np.uint32), out_o.view(np.uint32)), f"slice bitwise (no_tiles={no_tiles})"
    # whole path through phl_filter
    out2 = L.filter(s, exact=True).cpu().numpy()
    assert np.array_equal(out2.view(np.uint32), out_o.view(np.uint32))
    assert np.array_equal(L.filter(s, exact=True, no_tiles=True).cpu().numpy().view(np.uint32), out_o.view(np.uint32))
    assert scaled_err(L.slice(vb).cpu().numpy(), out_o) <= 1e-6      # default slice: one final multiply
    assert scaled_err(L.slice(vb, no_tiles=True).cpu().numpy(), out_o) <= 1e-6
    out2f = L.filter(s).cpu().numpy()               # default (fast) path
    assert scaled_err(out2f, out_o) <= 1e-5
    # contractual per-element tolerance on non-negative values (what mean-field feeds: probabilities)
    pos = np.abs(src)
    assert rel_err(L.filter(torch.from_numpy(pos).cuda()).cpu().numpy(), O.filter(pos)) <= RTOL
    # fused "- U" epilogue == LatticeGaussian (gaussian_matrix.py:303)
    out4 = L.filter(s, subtract_input=True, exact=True).cpu().numpy()
    assert np.array_equal(out4.view(np.uint32), (out_o - src).view(np.uint32))
    out5 = L.filter(s, subtract_input=True).cpu().numpy()
    assert np.abs(out5 - (out_o - src)).max() <= 1e-5 * np.abs(out_o).max()


def test_golden_lattice_vectors(phl, golden_dir):
    """Expected values here are the REFERENCE's own outputs (tests/golden/generate.py)."""
    files = sorted(glob.glob(os.path.join(golden_dir, "lattice_*.npz")))
    assert len(files) >= 7
    for f in files:
        g = np.load(f)
        L = phl.Lattice(torch.from_numpy(g["ref"]).cuda())
        assert L.M == int(g["M"]), f
        keys = L.keys()
        order = np.lexsort(keys.T[::-1])
        assert np.array_equal(keys[order], g["keys_sorted"]), f
        vid, w = L.replay()
        assert np.array_equal(keys[vid], g["replay_key"]), f
        assert np.array_equal(w.view(np.uint32), g["replay_w"].view(np.uint32)), f
        s = torch.from_numpy(g["src"]).cuda()
        vs = L.splat(s, exact=True)
        ft = lambda v: L.to_first_touch(v).cpu().numpy()
        assert np.array_equal(ft(vs)[order].view(np.uint32), g["splat_sorted"].view(np.uint32)), f
        assert scaled_err(ft(L.splat(s))[order], g["splat_sorted"]) <= 1e-5, f
        vb = L.blur(vs)
        assert np.array_equal(ft(vb)[order].view(np.uint32), g["blur_sorted"].view(np.uint32)), f
        out = phl.filter(s, torch.from_numpy(g["ref"]).cuda()).cpu().numpy()
        assert scaled_err(out, g["out"]) <= 1e-5, f
        out = L.filter(s, exact=True).cpu().numpy()
        assert np.array_equal(out.view(np.uint32), g["out"].view(np.uint32)), f


def test_strided_views_and_cpu_tensors(phl):
    """NCHW-style permuted views (gaussian_matrix.py:348-349) and CPU tensors in / out."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(5)
    h, w, Lc, d = 24, 40, 12, 5
    src_chw = rng.standard_normal((Lc, h, w)).astype(np.float32)
    ref_chw = (rng.random((d, h, w)) * 3).astype(np.float32)
    src_view = torch.from_numpy(src_chw).cuda().view(Lc, -1).permute(1, 0)   # [n, L], strides (1, n)
    ref_view = torch.from_numpy(ref_chw).cuda().view(d, -1).permute(1, 0)
    want = po.oracle_filter(np.ascontiguousarray(src_view.cpu().numpy()), np.ascontiguousarray(ref_view.cpu().numpy()))
    got = phl.filter(src_view, ref_view)
    assert got.is_cuda and scaled_err(got.cpu().numpy(), want) <= 1e-5
    # write into a permuted output
    Lat = phl.Lattice(ref_view)
    out_chw = torch.empty((Lc, h * w), device="cuda")
    Lat.filter(src_view, out=out_chw.permute(1, 0), exact=True)
    assert np.array_equal(out_chw.permute(1, 0).cpu().numpy().view(np.uint32), want.view(np.uint32))
    # row-padded pixel-major input
    padded = torch.zeros((h * w, Lc + 4), device="cuda")
    padded[:, :Lc] = src_view
    got2 = Lat.filter(padded[:, :Lc], exact=True)
    assert np.array_equal(got2.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert scaled_err(Lat.filter(padded[:, :Lc]).cpu().numpy(), want) <= 1e-5
    # CPU tensors are computed on the GPU and returned on the CPU (reference call shape)
    got3 = phl.filter(src_view.cpu(), ref_view.cpu())
    assert not got3.is_cuda and scaled_err(got3.numpy(), want) <= 1e-5


def test_error_behaviour(phl):
    ref = torch.rand(100, 3, device="cuda")
    with pytest.raises(AssertionError, match="Incompatible shapes"):
        phl.filter(torch.rand(99, 4, device="cuda"), ref)
    with pytest.raises(TypeError):
        phl.filter(torch.rand(100, 4, device="cuda", dtype=torch.float64), ref)
    with pytest.raises(phl.PhlError) as e:
        phl.Lattice(torch.rand(10, 17, device="cuda"))
    assert e.value.status == 7
    with pytest.raises(phl.PhlError) as e:   # int16 key overflow is reported, the reference wraps silently
        phl.Lattice(torch.rand(100, 2, device="cuda") * 1e5)
    assert e.value.status == 5
    # empty input
    L = phl.Lattice(torch.empty(0, 3, device="cuda"))
    assert L.M == 0 and L.filter(torch.empty(0, 4, device="cuda")).shape == (0, 4)


def test_cache_is_invisible(phl):
    from oracle import phl_oracle as po

    phl.clear_cache()
    ref = torch.rand(500, 3, device="cuda") * 4
    src = torch.randn(500, 6, device="cuda")
    a = phl.filter(src, ref)
    b = phl.filter(src, ref)          # cache hit
    assert torch.equal(a, b)
    ref.mul_(0.5)                     # in-place edit must invalidate
    c = phl.filter(src, ref)
    want = po.oracle_filter(src.cpu().numpy(), ref.cpu().numpy())
    assert scaled_err(c.cpu().numpy(), want) <= 1e-5


def test_tile_structures_image_like(phl):
    """LDS-staged path on an image-like feature set: chunks are ~16x16 pixel tiles, most
    vertices are fed by few chunks, and the staged kernels are selected."""
    from oracle import phl_oracle as po

    H, W, L = 96, 128, 32
    rng = np.random.default_rng(0)
    feat = np.empty((H, W, 5), np.float32)
    feat[..., 0] = (np.arange(W, dtype=np.float32) / 4)[None, :]
    feat[..., 1] = (np.arange(H, dtype=np.float32) / 4)[:, None]
    feat[..., 2:] = np.kron(rng.random((H // 8, W // 8, 3)).astype(np.float32), np.ones((8, 8, 1), np.float32)) * 5
    ref = feat.reshape(-1, 5)
    src = rng.random((H * W, L), dtype=np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    st = Lat.tile_stats(L)
    assert st["pixels_per_chunk"] == 256 and st["chunks"] == H * W // 256
    assert st["staged_splat"] == 1 and st["staged_slice"] == 1
    assert st["slots"] < 3 * H * W and st["multi_chunk_slots"] <= st["slots"]
    want = po.Oracle(ref).filter(src)
    got = Lat.filter(torch.from_numpy(src).cuda()).cpu().numpy()
    assert rel_err(got, want) <= 1e-5          # non-negative values: per-element relative
    # determinism of the staged path (no float atomics anywhere)
    got2 = Lat.filter(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(got, got2)


def test_poor_sharing_falls_back_to_gather(phl):
    """iid features: almost every (pixel, remainder) pair has its own vertex, so a chunk has
    ~(d+1)*P local vertices and staging them buys nothing: the host must pick the plain gather
    slice (results are the same either way)."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(3)
    n, d, L = 20000, 5, 16
    ref = (rng.random((n, d), dtype=np.float32) * 30).astype(np.float32)
    src = rng.random((n, L), dtype=np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    st = Lat.tile_stats(L)
    assert st["slots"] > 3 * n and st["staged_slice"] == 0
    want = po.Oracle(ref).filter(src)
    assert rel_err(Lat.filter(torch.from_numpy(src).cuda()).cpu().numpy(), want) <= 1e-5


@pytest.mark.parametrize("n,d,vd", [(5000, 5, 768), (777, 3, 36), (4099, 1, 8), (9000, 16, 12), (300, 5, 260),
                                    (40, 5, 256), (60, 3, 128), (100, 5, 128), (150, 2, 64), (33, 5, 100)])
def test_staged_path_shapes(phl, n, d, vd):
    """wide channel counts (several slabs), ragged last chunk, d = 1 and d = 16 through the
    default (LDS-staged where eligible) path; the tiny inputs take the 128- and 256-channel slabs
    (lane groups of 2 and 4 DPP rows in the segmented sums)."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(n)
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.05, axis=0).astype(np.float32)   # smooth: vertices are shared
    src = rng.random((n, vd), dtype=np.float32)
    O = po.Oracle(ref)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    want = O.filter(src)
    got = Lat.filter(torch.from_numpy(src).cuda()).cpu().numpy()
    assert rel_err(got, want) <= 1e-5
    exact = Lat.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
    assert np.array_equal(exact.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("workload", ["c3", "c2", "c5"])
def test_full_size_properties(phl, workload):
    """BASELINE.json's full-size volumes -- configs[2] 2048x1536x256, configs[1] 1390x1110x256 (ragged 16x16
    tiling: neither side is a multiple of the chunk edge) and configs[4] 1024x1024x128 (the 128-channel slab
    path at scale) -- are far beyond what the CPU oracle finishes in seconds, so they are checked through
    size-independent properties of the operator: symmetry <y, Wx> == <x, Wy> (the reference relies on it:
    gaussian_matrix.py:445-446), linearity, determinism, agreement of the default and reference-exact paths,
    W1 >= 0 -- and a crop of the same features against the CPU oracle."""
    sys_path_bench = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    __import__("sys").path.insert(0, sys_path_bench)
    import bench

    H, W, L, _ = bench.WORKLOADS[workload]
    feat = bench.synthetic_features(H, W)
    dev = torch.device("cuda")
    Lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    assert 0.02 < Lat.M / (H * W) < 0.5 and Lat.M >= 16383
    st = Lat.tile_stats(L)
    assert st["staged_splat"] == 1 and st["staged_slice"] == 1
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((H * W, L), device=dev, generator=g)
    y = torch.rand((H * W, L), device=dev, generator=g)
    Wx = Lat.filter(x)
    assert torch.equal(Wx, Lat.filter(x))                                   # deterministic
    Wy = Lat.filter(y)
    a, b = torch.sum(y.double() * Wx.double()), torch.sum(x.double() * Wy.double())
    # S^T (B_d ... B_0) S: symmetric up to the non-commutation of the per-axis blurs at missing
    # neighbours (the Jacobi passes run in a fixed axis order) and fp32 rounding; the reference treats it as
    # symmetric.  Measured 2.9e-6 on c3 (round 1); bound = 10x that.
    sym = abs(float(a - b)) / abs(float(a))
    Wxy = Lat.filter(x + 2 * y)
    lin = float((Wxy - (Wx + 2 * Wy)).abs().max()) / float(Wxy.abs().max())
    Wx_exact = Lat.filter(x, exact=True)
    ex = float(((Wx - Wx_exact).abs() / Wx_exact.abs().clamp_min(1e-3 * float(Wx_exact.max()))).max())
    print(f"[measured] {workload}: M/n={Lat.M / (H * W):.4f} symmetry {sym:.2e} linearity {lin:.2e} default-vs-exact {ex:.2e}")
    assert sym <= 3e-5
    assert lin <= 5e-6            # fp32 rounding of two different summation groupings
    assert ex <= 1e-5             # default path vs the reference-exact arithmetic, per element (<< north star's 1e-4)
    del Wxy, Wx_exact, Wy, y
    deg = Lat.filter(torch.ones((H * W, 4), device=dev))
    assert float(deg.min()) > 0 and torch.equal(deg[:, 0], deg[:, 3])
    # crops of the volume against the CPU oracle on features that share the lattice scale: top-left, and the
    # bottom-right corner (ragged last tiles for c2)
    from oracle import phl_oracle as po
    for (r0, c0) in ((0, 0), (H - 96, W - 128)):
        crop = np.ascontiguousarray(feat[r0:r0 + 96, c0:c0 + 128].reshape(-1, 5))
        xs = x[:96 * 128, :8].cpu().numpy().copy()
        want = po.Oracle(crop).filter(xs)
        cl = phl.Lattice(torch.from_numpy(crop).to(dev))
        got = cl.filter(torch.from_numpy(xs).to(dev)).cpu().numpy()
        assert rel_err(got, want) <= 1e-5
        gote = cl.filter(torch.from_numpy(xs).to(dev), exact=True).cpu().numpy()
        assert np.array_equal(gote.view(np.uint32), want.view(np.uint32))


def test_wide_crop_of_c2_against_oracle(phl):
    """A 256-channel, ragged-size crop of configs[1]'s features against the CPU oracle: the full channel
    count through every slab of the staged kernels, lattice above the reference's first table doubling."""
    sys_path_bench = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    __import__("sys").path.insert(0, sys_path_bench)
    import bench
    from oracle import phl_oracle as po

    H, W, L, _ = bench.WORKLOADS["c2"]
    feat = bench.synthetic_features(H, W, sigma_xy=3.0)          # sigma_xy = 3 px: M/n ~ 0.3, more vertices per crop
    crop = np.ascontiguousarray(feat[:250, :330].reshape(-1, 5))   # 250 x 330: not multiples of the chunk edge
    rng = np.random.default_rng(8)
    src = rng.random((crop.shape[0], L), dtype=np.float32)
    O = po.Oracle(crop)
    assert O.M >= 16383
    want = O.filter(src)
    Lat = phl.Lattice(torch.from_numpy(crop).cuda())
    assert Lat.M == O.M
    s = torch.from_numpy(src).cuda()
    assert np.array_equal(Lat.filter(s, exact=True).cpu().numpy().view(np.uint32), want.view(np.uint32))
    e = rel_err(Lat.filter(s).cpu().numpy(), want)
    print(f"[measured] c2 crop 330x250x256, M={O.M}: default path vs oracle, per-element relative {e:.2e}")
    assert e <= 1e-5


@pytest.mark.parametrize("vd", [1, 3])
def test_narrow_values_at_c2_geometry(phl, vd):
    """Value tensors that are not a multiple of four channels wide -- the degree vector of RbfLaplacian (vd = 1,
    crf/gaussian_matrix.py:311-312) and RGB (vd = 3, crf/lattice/lite/test_bilateral.ipynb) -- on configs[1]'s geometry:
    the full-size call for determinism and timing, a 250 x 330 crop against the CPU oracle (bit-identical in exact
    mode).  These widths run on the scalar-lane gather kernels (the staged kernels move 16-byte pieces)."""
    import time

    sys_path_bench = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    __import__("sys").path.insert(0, sys_path_bench)
    import bench
    from oracle import phl_oracle as po

    H, W, _, _ = bench.WORKLOADS["c2"]
    feat = bench.synthetic_features(H, W)
    dev = torch.device("cuda")
    Lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True)
    x = torch.rand((H * W, vd), device=dev, generator=torch.Generator(device=dev).manual_seed(vd))
    a = Lat.filter(x)
    assert torch.equal(a, Lat.filter(x)) and torch.isfinite(a).all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        Lat.filter(x, out=a)
    torch.cuda.synchronize()
    print(f"[measured] c2 geometry, vd = {vd}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per filter call")
    crop = np.ascontiguousarray(feat[H - 250:, W - 330:].reshape(-1, 5))
    xs = np.random.default_rng(vd).random((crop.shape[0], vd), dtype=np.float32)
    want = po.Oracle(crop, faithful_table=True).filter(xs)
    cl = phl.Lattice(torch.from_numpy(crop).to(dev), reference_table=True)
    got = cl.filter(torch.from_numpy(xs).to(dev)).cpu().numpy()
    assert rel_err(got, want) <= 1e-5
    gote = cl.filter(torch.from_numpy(xs).to(dev), exact=True).cpu().numpy()
    assert np.array_equal(gote.view(np.uint32), want.view(np.uint32))


GROWTH = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "growth_*.npz")))


@pytest.mark.parametrize("path", GROWTH, ids=os.path.basename)
def test_reference_output_above_table_doubling(phl, path):
    """Stored outputs of the REFERENCE ENGINE at M >= 16383 (its hash table has doubled; every BASELINE GPU
    config is in this regime).  `defect_mask` marks the rows reached by the reference's stale-slot-after-grow
    defect (permutohedral.h:59-62,101-103: lookup() hashes before grow(); see oracle/phl_oracle.c).
      * default build (defect-free table): exact arithmetic is BIT-EQUAL to the reference outside the mask and
        to the clean oracle everywhere; the default arithmetic is within 1e-4 per element outside the mask;
      * reference_table=True build: reproduces the reference's table, duplicates included -- bit-equal to the
        reference's output on EVERY row."""
    from _golden_util import load_growth_case
    from oracle import phl_oracle as po

    g = load_growth_case(path)
    mask = g["mask"]
    ref = torch.from_numpy(g["ref"]).cuda()
    s = torch.from_numpy(g["src"]).cuda()
    L = phl.Lattice(ref)
    assert L.M == g["clean_M"]
    out = L.filter(s, exact=True).cpu().numpy()
    assert np.array_equal(out[~mask].view(np.uint32), g["out"][~mask].view(np.uint32))
    clean = po.Oracle(g["ref"]).filter(g["src"])
    assert np.array_equal(out.view(np.uint32), clean.view(np.uint32))
    fast = L.filter(s).cpu().numpy()
    e = rel_err(fast[~mask], g["out"][~mask])
    print(f"[measured] {os.path.basename(path)}: M={g['M']}, defect mask {int(mask.sum())}/{len(mask)} rows = "
          f"{mask.mean():.3%}; default path outside the mask: {e:.2e} relative")
    assert e <= 1e-5 if (g["src"] >= 0).all() else scaled_err(fast[~mask], g["out"][~mask]) <= 1e-5
    assert 0 < mask.mean() < 0.05
    # the opt-in reference table: identical to the reference everywhere
    Lr = phl.Lattice(ref, reference_table=True)
    assert Lr.M == g["M"]
    outr = Lr.filter(s, exact=True).cpu().numpy()
    assert np.array_equal(outr.view(np.uint32), g["out"].view(np.uint32))
    Of = po.Oracle(g["ref"], faithful_table=True)
    assert np.array_equal(Lr.keys(), Of.keys())
    vid, w = Lr.replay()
    ovid, ow = Of.replay()
    assert np.array_equal(vid, ovid) and np.array_equal(Lr.neighbors(), Of.neighbors())
    fr = Lr.filter(s).cpu().numpy()
    assert (rel_err(fr, g["out"]) <= 1e-5) if (g["src"] >= 0).all() else (scaled_err(fr, g["out"]) <= 1e-5)


@pytest.mark.parametrize("n,d,vd,scale,seed", [(20000, 5, 3, 8.0, 20000), (40000, 3, 2, 30.0, 40000), (30000, 2, 8, 300.0, 6),
                                                (50000, 4, 4, 12.0, 8), (200000, 3, 2, 40.0, 23)])
def test_reference_table_mode_equals_the_reference_engine(phl, n, d, vd, scale, seed):
    """Lattice(reference_table=True) against the oracle's faithful-table mode (== the reference engine, bit for
    bit, incl. its duplicate vertices above M = 16383): numbering, replay, blur neighbours, every stage."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(seed)
    ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    src = rng.standard_normal((n, vd)).astype(np.float32)
    O = po.Oracle(ref, faithful_table=True)
    out_o, splat_o, blur_o = O.filter(src, stages=True)
    L = phl.Lattice(torch.from_numpy(ref).cuda(), reference_table=True)
    assert L.M == O.M and L.M >= 16383
    assert np.array_equal(L.keys(), O.keys())
    vid, w = L.replay()
    ovid, ow = O.replay()
    assert np.array_equal(vid, ovid) and np.array_equal(w.view(np.uint32), ow.view(np.uint32))
    assert np.array_equal(L.neighbors(), O.neighbors())
    s = torch.from_numpy(src).cuda()
    vs = L.splat(s, exact=True)
    assert np.array_equal(L.to_first_touch(vs).cpu().numpy().view(np.uint32), splat_o.view(np.uint32))
    vb = L.blur(vs)
    assert np.array_equal(L.to_first_touch(vb).cpu().numpy().view(np.uint32), blur_o.view(np.uint32))
    assert np.array_equal(L.filter(s, exact=True).cpu().numpy().view(np.uint32), out_o.view(np.uint32))
    assert scaled_err(L.filter(s).cpu().numpy(), out_o) <= 1e-5
    if po.reference_available():          # the reference engine itself, when its binary travelled with the tree
        R = po.reference_filter(src, ref)
        assert np.array_equal(L.filter(s, exact=True).cpu().numpy().view(np.uint32), R.view(np.uint32))


def test_reference_table_doubling_inside_blur(phl):
    """M == 2^14 - 1 when splat ends: the reference's table doubles inside blur()'s first neighbour lookup, which
    is then probed from a stale slot (permutohedral.h:62 has no `create` test)."""
    from oracle import phl_oracle as po

    for seed in range(30):
        rng = np.random.default_rng(99 + seed)
        ref = (rng.random((12000, 5), dtype=np.float32) * np.float32(9.0)).astype(np.float32)
        vid = po.Oracle(ref).replay()[0]
        first = int(np.nonzero(vid.max(1) >= 16382)[0][0])
        if vid[first].max() == 16382:
            break
    else:
        pytest.fail("no prefix with exactly 16383 vertices")
    ref = np.ascontiguousarray(ref[:first + 1])
    O = po.Oracle(ref, faithful_table=True)
    assert O.M == 16383
    L = phl.Lattice(torch.from_numpy(ref).cuda(), reference_table=True)
    assert L.M == O.M and np.array_equal(L.neighbors(), O.neighbors())
    src = rng.standard_normal((ref.shape[0], 4)).astype(np.float32)
    got = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.filter(src).view(np.uint32))


def test_randomised_shapes_against_oracle(phl):
    """40 random (n, d, vd, feature scale, layout) draws: exact mode bit-identical to the oracle,
    default mode within fp32 rounding; covers tiny n, every LPR / slab width, smooth and iid
    features, padded and offset rows."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(20261004)
    for trial in range(40):
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 257, 1000, 4097, 20011]))
        d = int(rng.integers(1, 9))
        vd = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 20, 32, 60, 64, 68, 128, 132, 256, 300]))
        if n * vd > 3_000_000:
            vd = 8
        scale = float(rng.choice([0.0, 0.3, 2.0, 8.0, 40.0]))
        smooth = bool(rng.integers(0, 2))
        ref = rng.random((n, d), dtype=np.float32) * np.float32(scale)
        if smooth:
            ref = np.cumsum(ref * np.float32(0.02), axis=0).astype(np.float32)
        src = rng.random((n, vd), dtype=np.float32) - np.float32(0.25)
        O = po.Oracle(ref)
        if O.status == 1:        # a lattice coordinate left int16: both sides must say so
            with pytest.raises(phl.PhlError) as err:
                phl.Lattice(torch.from_numpy(ref).cuda())
            assert err.value.status == 5
            continue
        want = O.filter(src)
        L = phl.Lattice(torch.from_numpy(ref).cuda())
        assert L.M == O.M, (trial, n, d, vd)
        pad = int(rng.choice([0, 4, 3]))
        buf = torch.zeros((n, vd + pad), device="cuda")
        buf[:, :vd] = torch.from_numpy(src).cuda()
        s = buf[:, :vd]
        exact = L.filter(s, exact=True).cpu().numpy()
        assert np.array_equal(exact.view(np.uint32), want.view(np.uint32)), (trial, n, d, vd, scale, smooth, pad)
        fast = L.filter(s).cpu().numpy()
        assert scaled_err(fast, want) <= 1e-5, (trial, n, d, vd, scale, smooth, pad)
        sub = L.filter(s, subtract_input=True).cpu().numpy()
        assert scaled_err(sub + src, want) <= 1e-5, (trial, n, d, vd)


def float64_splat_truth(O, src):
    """The filter with the splat's sums taken in float64 over the oracle's own (vertex, weight) entries, then the
    oracle's blur and slice (sums of three and of d+1 terms: 1e-7): what both fp32 summation orders approximate."""
    vid, w = O.replay()
    V = np.zeros((O.M, src.shape[1]))
    s64 = src.astype(np.float64)
    for k in range(vid.shape[1]):
        np.add.at(V, vid[:, k], w[:, k].astype(np.float64)[:, None] * s64)
    return O.slice(O.blur(V.astype(np.float32)))


def test_long_segments_default_arithmetic_is_nearer_float64_than_the_reference(phl):
    """Found by tools/fuzz_filter.py (seed 9106, trial 6): constant features put all 50021 pixels on d+1 vertices.
    The reference sums a vertex's pixels one after the other in fp32 (permutohedral.h:441-447) and carries that chain's
    rounding -- 1.2e-5 of the largest output here; the default arithmetic sums per chunk and then across chunks, so it
    sits at that distance from the reference and an order of magnitude nearer the float64 sums.  Exact mode reproduces
    the reference's chain bit for bit."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(9106)
    n, d, vd = 50021, 3, 64
    ref = np.zeros((n, d), np.float32)
    src = rng.random((n, vd), dtype=np.float32) - np.float32(0.25)
    O = po.Oracle(ref)
    want = O.filter(src)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    s = torch.from_numpy(src).cuda()
    assert L.M == O.M == d + 1
    assert np.array_equal(L.filter(s, exact=True).cpu().numpy().view(np.uint32), want.view(np.uint32))
    fast = L.filter(s).cpu().numpy()
    truth = float64_splat_truth(O, src)
    e_ref, e_fast, dist = scaled_err(want, truth), scaled_err(fast, truth), scaled_err(fast, want)
    print(f"[measured] {n} pixels on {d + 1} vertices: reference vs float64 sums {e_ref:.2e}, default vs float64 sums {e_fast:.2e}, "
          f"default vs reference {dist:.2e}")
    assert e_fast <= 2e-6 and e_fast <= e_ref
    assert dist <= e_ref + e_fast + 1e-7 and dist <= RTOL


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6])
def test_fused_blur_equals_axis_by_axis(phl, d):
    """phl_blur takes the axes two per pass (k_blur2) and a last single one when d+1 is odd: same bits as
    d+1 phl_blur_axis calls, and as the CPU restatement."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(20 + d)
    n, vd = 6000, 24
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.05, axis=0).astype(np.float32)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    v0 = torch.from_numpy(rng.standard_normal((L.M, vd)).astype(np.float32)).cuda()
    a, b = v0.clone(), torch.empty_like(v0)
    for axis in range(d + 1):
        L.blur_axis(axis, a, b)
        a, b = b, a
    fused = L.blur(v0.clone())
    assert torch.equal(fused, a)
    want = po.Oracle(ref).blur(L.to_first_touch(v0).cpu().numpy())     # the oracle numbers vertices in first-touch order
    assert np.array_equal(L.to_first_touch(fused).cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert torch.equal(L.from_first_touch(L.to_first_touch(v0)), v0)


def test_more_than_two_million_vertices(phl):
    """M >= 2^21 switches the chunk sort to 64-bit keys, and pixels that share nothing overflow the
    fixed-stride slot scratch, so the chunk pass is repeated with exact offsets: both rare paths, checked
    through the slice (which uses the chunk structures whenever they fit) against the CPU restatement."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(77)
    n, d, vd = 380_000, 5, 4
    ref = (rng.random((n, d), dtype=np.float32) * np.float32(400.0)).astype(np.float32)
    src = rng.random((n, vd), dtype=np.float32)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    assert L.M >= (1 << 21)
    st = L.tile_stats(vd)
    assert st["slots"] == n * (d + 1) and st["max_local_vertices"] > 384
    O = po.Oracle(ref)
    assert O.M == L.M
    want = O.filter(src)
    assert rel_err(L.filter(torch.from_numpy(src).cuda()).cpu().numpy(), want) <= 1e-5
    vb = L.blur(L.splat(torch.from_numpy(src).cuda(), exact=True))
    for no_tiles in (False, True):
        out = L.slice(vb.clone(), exact=True, no_tiles=no_tiles).cpu().numpy()
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), no_tiles


def test_chunk_grouping_with_one_to_three_digit_passes(phl):
    """k_chunk_group sorts a chunk's entries by dense local vertex id, four bits a pass: chunks with 1, <= 16, <= 256
    and > 256 distinct vertices (0 ... 3 passes) in the LDS-staged filter against the CPU restatement."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(3)
    n = 4096
    for name, ref in (
        ("one simplex", np.full((n, 5), 0.37, np.float32) + rng.random((n, 5), dtype=np.float32) * 1e-4),
        ("a few vertices", np.cumsum(rng.random((n, 5), dtype=np.float32) * 0.002, axis=0).astype(np.float32)),
        ("image-like", np.cumsum(rng.random((n, 5), dtype=np.float32) * 0.03, axis=0).astype(np.float32)),
        ("iid", (rng.random((n, 5), dtype=np.float32) * 40).astype(np.float32)),
    ):
        src = rng.random((n, 16), dtype=np.float32)
        L = phl.Lattice(torch.from_numpy(ref).cuda())
        st = L.tile_stats(16)
        want = po.Oracle(ref).filter(src)
        got = L.filter(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), (name, st)
        exact = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
        assert np.array_equal(exact.view(np.uint32), want.view(np.uint32)), (name, st)
        print(name, st)


def test_filter_is_graph_capturable(phl):
    """After phl_reserve the filter launch sequence allocates nothing and never synchronises, so
    it can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed on new input values."""
    rng = np.random.default_rng(9)
    n, d, L = 20000, 5, 32
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.02, axis=0).astype(np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    Lat.reserve(L)
    x = torch.rand((n, L), device="cuda")
    out = torch.empty_like(x)
    Lat.filter(x, out=out)                       # warm-up outside the capture
    want1 = out.clone()
    torch.cuda.synchronize()
    dead = phl.Lattice(torch.from_numpy(ref[:500]).cuda())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Lat.filter(x, out=out)
        del dead                                  # a lattice dying mid-capture (gc) must not invalidate it
    x.copy_(torch.rand((n, L), device="cuda"))   # new values in the captured input buffer
    g.replay()
    torch.cuda.synchronize()
    want2 = Lat.filter(x)
    assert torch.equal(out, want2) and not torch.equal(out, want1)


@pytest.mark.parametrize("layout", ["pixel_major", "nchw_view", "exact"])
def test_reserve_then_capture_without_warm_up(phl, layout):
    """phl_reserve's contract (include/phl.h): after it the NEXT filter call allocates nothing -- so a capture
    may follow a single reserve() directly, with no un-captured warm-up call, in the pixel-major layout, through
    the channel-major views BatchedAdjacency passes (gaussian_matrix.py:348-349) and for exact-mode calls."""
    rng = np.random.default_rng(19)
    n, d, L = 30000, 5, 32
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.02, axis=0).astype(np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    exact = layout == "exact"
    Lat.reserve(L, strided_io=(layout == "nchw_view"), exact=exact)
    if layout == "nchw_view":
        x = torch.rand((L, n), device="cuda").permute(1, 0)
        out = torch.empty((L, n), device="cuda").permute(1, 0)
    else:
        x = torch.rand((n, L), device="cuda")
        out = torch.empty_like(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Lat.filter(x, out=out, exact=exact)
    g.replay()
    torch.cuda.synchronize()
    want = phl.Lattice(torch.from_numpy(ref).cuda()).filter(x.contiguous(), exact=exact)
    assert torch.equal(out.contiguous(), want)


def test_two_threads_two_streams_one_lattice(phl):
    """SURVEY 8(b) threading row: "re-entrant; autograd may call from any thread".  Two host threads, each on its
    own stream, filter different values through ONE cached lattice at the same time; every result must equal
    the single-threaded one bit for bit (each call works on its own value workspace)."""
    import threading

    rng = np.random.default_rng(23)
    n, d, L = 120000, 5, 64
    ref_np = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.01, axis=0).astype(np.float32)
    ref = torch.from_numpy(ref_np).cuda()
    phl.clear_cache()
    xs = [torch.rand((n, L), device="cuda") for _ in range(4)]
    want = [phl.filter(x, ref) for x in xs]
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(25):
                    k = (tid + it) % len(xs)
                    got = phl.filter(xs[k], ref)             # same cached Lattice in both threads
                    if it % 5 == 4:
                        st.synchronize()
                        if not torch.equal(got, want[k]):
                            errors.append((tid, it, float((got - want[k]).abs().max())))
                st.synchronize()
                if not torch.equal(got, want[k]):
                    errors.append((tid, "last", float((got - want[k]).abs().max())))
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert phl.lattice_for(ref) is phl.lattice_for(ref)


@pytest.mark.parametrize("kind", ["constant", "half_flat_colour", "outliers"])
def test_degenerate_features_build_in_bounded_time(phl, kind):
    """Clustered features: a constant `ref` puts every pixel into one grid cell and one vertex list, a colour-only
    ref with a flat background half of them.  The build's grouping steps are O(n) radix sorts, so this takes
    milliseconds (a rank sort per list would take minutes here and look like a hang)."""
    import time
    from oracle import phl_oracle as po

    n, L = 1 << 20, 8
    rng = np.random.default_rng(31)
    if kind == "constant":
        ref = np.full((n, 5), 0.37, np.float32)
    elif kind == "half_flat_colour":
        ref = (rng.random((n, 3), dtype=np.float32) * 4).astype(np.float32)
        ref[: n // 2] = np.float32(1.25)                    # flat background
    else:
        ref = (rng.random((n, 2), dtype=np.float32) * 0.5).astype(np.float32)
        ref[::100000] += np.float32(3000.0)                 # range outliers: the uniform grid collapses
    src = rng.random((n, L), dtype=np.float32)
    r = torch.from_numpy(ref).cuda()
    s = torch.from_numpy(src).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    Lat = phl.Lattice(r)
    out = Lat.filter(s)
    torch.cuda.synchronize()
    t_default = time.time() - t0
    t0 = time.time()
    oute = Lat.filter(s, exact=True)                        # builds the pixel-sorted lists: one list of n entries
    torch.cuda.synchronize()
    t_exact = time.time() - t0
    print(f"[measured] {kind}: n={n} M={Lat.M} build+filter {t_default * 1e3:.1f} ms, exact (lists + filter) {t_exact * 1e3:.1f} ms")
    assert t_default < 5.0 and t_exact < 20.0
    cut = 40000                                             # oracle on a prefix (same clustering)
    O = po.Oracle(ref[:cut])
    Lc = phl.Lattice(r[:cut].contiguous())
    assert Lc.M == O.M
    want = O.filter(src[:cut])
    assert np.array_equal(Lc.filter(s[:cut].contiguous(), exact=True).cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert rel_err(Lc.filter(s[:cut].contiguous()).cpu().numpy(), want) <= 1e-4
    # full size: default (chunk partial sums) vs exact (the reference's sequential fp32 sum of up to 2^20 terms per
    # vertex, whose own rounding error is ~sqrt(n) ulp): only a sanity bound here
    assert torch.isfinite(out).all() and rel_err(out.cpu().numpy(), oute.cpu().numpy()) <= 5e-3


def test_very_wide_values_many_slabs(phl):
    """vd = 3072 (the reference's ref-gradient filter for L=256, d=5 has 2L(1+d) channels): 48
    slabs through the staged kernels."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(12)
    n, d, vd = 600, 5, 3072
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.03, axis=0).astype(np.float32)
    src = rng.random((n, vd), dtype=np.float32)
    want = po.Oracle(ref).filter(src)
    got = phl.Lattice(torch.from_numpy(ref).cuda()).filter(torch.from_numpy(src).cuda()).cpu().numpy()
    assert rel_err(got, want) <= 1e-5


def test_block_cache_is_invisible(phl):
    """Lattice arrays come from a device block cache (csrc/phl_api.hip): building, destroying and rebuilding lattices
    of the same and of different sizes reuses blocks without changing any result; PHL_CACHE_MAX_MB=0 (fresh
    process) gives the same bits with plain hipMalloc / hipFree; phl_trim_scratch releases everything."""
    import subprocess
    import sys
    from oracle import phl_oracle as po

    rng = np.random.default_rng(41)
    outs = []
    for n in (30000, 30000, 12000, 30000, 50000):
        ref = np.cumsum(rng.random((n, 5), dtype=np.float32) * 0.02, axis=0).astype(np.float32)
        src = rng.random((n, 16), dtype=np.float32)
        L = phl.Lattice(torch.from_numpy(ref).cuda())
        got = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), po.Oracle(ref).filter(src).view(np.uint32)), n
        outs.append(got)
        L.close()                                           # its arrays go to the cache; the next build takes them
    assert phl.load_library().phl_trim_scratch() == 0
    code = (
        "import os, sys, numpy as np, torch\n"
        "root = sys.argv[1]\n"
        "sys.path[:0] = [os.path.join(root, 'depth-estimation_amd'), root]\n"
        "import phl\n"
        "rng = np.random.default_rng(41)\n"
        "ref = np.cumsum(rng.random((30000, 5), dtype=np.float32) * 0.02, axis=0).astype(np.float32)\n"
        "src = rng.random((30000, 16), dtype=np.float32)\n"
        "for _ in range(3):\n"
        "    L = phl.Lattice(torch.from_numpy(ref).cuda()); out = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy(); L.close()\n"
        "np.save(sys.argv[2], out)\n")
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "o.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code, root, path], env=dict(os.environ, PHL_CACHE_MAX_MB="0"),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.array_equal(np.load(path).view(np.uint32), outs[0].view(np.uint32))


def test_threads_with_different_channel_counts_share_one_lattice(phl):
    """Workspace pool under stress: three host threads, each on its own stream, filter values of DIFFERENT widths
    (so workspaces are grown and recycled while others are in flight), with and without the fused subtraction,
    through one lattice; every result equals the single-threaded one bit for bit."""
    import threading

    rng = np.random.default_rng(77)
    n, d = 60000, 5
    ref_np = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.015, axis=0).astype(np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref_np).cuda())
    widths = [4, 16, 64, 100, 256, 3]
    xs = {vd: torch.rand((n, vd), device="cuda") for vd in widths}
    want = {(vd, sub): Lat.filter(xs[vd], subtract_input=sub) for vd in widths for sub in (False, True)}
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            r = np.random.default_rng(100 + tid)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(40):
                    vd = int(r.choice(widths))
                    sub = bool(r.integers(0, 2))
                    got = Lat.filter(xs[vd], subtract_input=sub)
                    if it % 4 == 3:
                        st.synchronize()
                        if not torch.equal(got, want[(vd, sub)]):
                            errors.append((tid, it, vd, sub, float((got - want[(vd, sub)]).abs().max())))
                st.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_analytic_and_simulated_table_replay_build_the_same_lattice(phl, monkeypatch):
    """PHL_BUILD_REFERENCE_TABLE through the analytic replay (default; its occupancy check runs on the device) and
    through the table simulation (PHL_REPLAY_FAST=0, also the fallback): identical vertices, per-pixel lookups,
    blur neighbours -- on shared-vertex features with several doublings and on iid features."""
    rng = np.random.default_rng(77)
    for n, d, scale in ((120000, 5, 9.0), (60000, 3, 30.0), (40000, 5, 60.0)):
        ref = torch.from_numpy((rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)).cuda()
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("PHL_REPLAY_FAST", mode)
            L = phl.Lattice(ref, reference_table=True)
            got[mode] = (L.M, L.keys(), L.replay()[0], L.neighbors())
        assert got["1"][0] == got["0"][0] >= 16383
        for a, b in zip(got["1"][1:], got["0"][1:]):
            assert np.array_equal(a, b)


def test_occupancy_check_on_the_device_equals_the_host(phl):
    """k_home_hist / k_add_homes / k_cluster_check (the analytic replay's one assumption, checked during the build)
    against the host form, which tests/test_reference_table_cpu.py holds against brute-force probing."""
    from test_reference_table_cpu import keys_with_homes, probe_paths, ref_homes

    rng = np.random.default_rng(23)
    seen = set()
    for trial in range(16):
        cap = 1 << int(rng.choice([15, 16, 17, 19]))
        keys, homes = keys_with_homes(rng, cap, count=1 << 19)
        n = int(rng.integers(cap // 8, min(cap // 2 - 1, len(keys) - 1)))
        if trial % 2:
            order = np.argsort(-homes, kind="stable")
            pile = order[: int(rng.integers(2, 300))]
            rest = rng.permutation(np.setdiff1d(np.arange(len(keys)), pile))[: n - len(pile)]
            sel = np.concatenate([rest, pile])
        else:
            sel = rng.permutation(len(keys))[:n]
        k, hk = keys[sel], homes[sel]
        extra = rng.integers(0, n, int(rng.integers(0, 4)))
        stale = rng.integers(0, n, int(rng.integers(0, 3)))
        for c in list(rng.integers(0, n, 2)) + [n - 1, int(np.argmax(hk))]:
            want = probe_paths(k, n, extra, stale, cap, [c], on_device=0)
            got = probe_paths(k, n, extra, stale, cap, [c], on_device=1)
            assert got == want, (trial, cap, n, c)
            seen.add(want)
        # several keys in one call: all must pass
        many = rng.integers(0, n, 40)
        assert probe_paths(k, n, extra, stale, cap, many, on_device=1) == probe_paths(k, n, extra, stale, cap, many, on_device=0)
    assert seen == {0, 1}, "the cases should cover both answers"
    # above 2^22 slots one workgroup does the whole table (k_cluster_check)
    cap = 1 << 23
    keys, homes = keys_with_homes(rng, cap, count=1 << 19)
    order = np.argsort(-homes, kind="stable")
    for pile_n in (0, 3):
        pile = order[:pile_n]
        sel = np.concatenate([rng.permutation(np.setdiff1d(np.arange(len(keys)), pile))[:300000], pile])
        # pile the last slots full by asking for the same homes again (extra entries)
        extra = np.repeat(np.arange(len(sel) - pile_n, len(sel)), 40) if pile_n else np.zeros(0, np.int64)
        for c in (0, len(sel) - 1):
            want = probe_paths(keys[sel], len(sel), extra, [], cap, [c], on_device=0)
            assert probe_paths(keys[sel], len(sel), extra, [], cap, [c], on_device=1) == want, (pile_n, c)


def test_concurrent_builds_from_threads(phl):
    """Several host threads build (reference-table) lattices and filter through them at the same time, each on its own
    stream: the build's per-thread state (pinned read-back arena, side stream of the pixel order), the shared block
    cache and the one scratch block (a second builder falls back to plain allocations) must not get in each other's
    way -- every result bit-identical to the same build done alone."""
    import threading

    import bench

    H, W, L = 256, 384, 16
    refs = [torch.from_numpy(bench.synthetic_features(H, W, sigma_xy=sx).reshape(-1, 5)).cuda() for sx in (8.0, 5.0, 3.0)]
    src = torch.rand((H * W, L), device="cuda")
    alone = []
    for r in refs:
        lat = phl.Lattice(r, reference_table=True)
        alone.append((lat.M, lat.filter(src).cpu().numpy()))
        lat.close()
    assert max(m for m, _ in alone) > 16383, "at least one lattice should go through the table replay"
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(6):
                    lat = phl.Lattice(refs[i], reference_table=True)
                    out = lat.filter(src)
                    st.synchronize()
                    if lat.M != alone[i][0] or not np.array_equal(out.cpu().numpy().view(np.uint32), alone[i][1].view(np.uint32)):
                        errs.append((i, it))
                    lat.close()
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(refs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs


def test_side_stream_of_the_table_replay_is_kept_per_device(phl):
    """The reference-table build launches the chunk build's pixel order on a side stream under the host's table replay.
    A thread keeps ONE such stream per device it builds on (a thread that deals batch items over several GPUs used to
    drop and re-create it at every device switch: one leaked stream and two events per build).  On one GPU: any number
    of builds hold one stream; with more than one GPU visible, builds alternating between two devices hold two."""
    import ctypes
    import threading

    import bench

    lib = phl.load_library()
    lib.phl_debug_side_streams.restype = ctypes.c_int
    feat = bench.synthetic_features(192, 512, sigma_xy=3.0).reshape(-1, 5)
    ndev = torch.cuda.device_count()
    seen = {}

    def work():
        devs = [torch.device("cuda", i) for i in range(min(ndev, 2))]
        for it in range(6):
            dev = devs[it % len(devs)]
            lat = phl.Lattice(torch.from_numpy(feat).to(dev), reference_table=True)
            assert lat.M > 16383                      # goes through the replay
            lat.close()
        seen["count"] = lib.phl_debug_side_streams()
        seen["devs"] = len(devs)

    t = threading.Thread(target=work)                 # a fresh thread: its own thread-local table
    t.start()
    t.join()
    assert seen["count"] == seen["devs"], seen


@pytest.mark.parametrize("vd", [9, 10, 30, 50, 130, 231])
def test_widths_and_rows_off_the_sixteen_byte_grid(phl, vd):
    """Value tensors the chunk kernels cannot take as they are -- a channel count that is not a multiple of 4 (the reference's
    max_disp = w // 6: 231 at 1390 columns), rows whose stride or base address is off the 16-byte grid (a column slice of a
    wider tensor) -- are, from 128 channels on, staged into 16-byte rows and run on the chunk kernels at the width rounded up
    (phl_filter; narrower ones keep the gather kernels, which win there): against the CPU oracle, with and without the fused subtraction, into an unaligned
    output view, and equal to the gather kernels' result within rounding."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(100 + vd)
    h, w, d = 40, 56, 5
    n = h * w
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    ref = np.stack([yy / 5, xx / 5] + [rng.random((h, w), dtype=np.float32) * 2 for _ in range(d - 2)], axis=-1).reshape(n, d)
    src = rng.standard_normal((n, vd)).astype(np.float32)
    want = po.oracle_filter(src, ref)
    dev = torch.device("cuda")
    Lat = phl.Lattice(torch.from_numpy(ref).to(dev))
    x = torch.from_numpy(src).to(dev)
    got = Lat.filter(x)
    assert scaled_err(got.cpu().numpy(), want) <= 1e-5
    assert scaled_err(Lat.filter(x, no_tiles=True).cpu().numpy(), want) <= 1e-5
    assert np.array_equal(Lat.filter(x, exact=True).cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert scaled_err(Lat.filter(x, subtract_input=True).cpu().numpy(), want - src) <= 1e-5
    # a column slice of a wider tensor (base address 4 bytes off the grid), and an output view of the same kind
    wide = torch.zeros((n, vd + 3), device=dev)
    wide[:, 1:vd + 1] = x
    out_wide = torch.full((n, vd + 5), -3.0, device=dev)
    res = Lat.filter(wide[:, 1:vd + 1], out=out_wide[:, 2:vd + 2])
    assert scaled_err(res.cpu().numpy(), want) <= 1e-5
    assert float(out_wide[:, :2].min()) == -3.0 and float(out_wide[:, vd + 2:].max()) == -3.0      # nothing written beside the view
    # repeatable (the staging buffers' extra channels never reach the result)
    assert torch.equal(Lat.filter(x), got)
    if vd % 4 == 2:       # a multiple of 4 channels inside a misaligned slice
        v4 = vd - 2
        res4 = Lat.filter(wide[:, 1:v4 + 1])
        assert scaled_err(res4.cpu().numpy(), want[:, :v4]) <= 1e-5


@pytest.mark.parametrize("layout", ["odd_width", "column_slice"])
def test_reserve_then_capture_rows_off_the_grid(phl, layout):
    """phl_reserve's contract for the widths phl_filter stages: 231 channels (not a multiple of 4: sized by reserve() itself)
    and a 160-channel column slice of a wider tensor (rows off the 16-byte grid: reserve(strided_io=True)) are captured
    directly behind one reserve() call, no warm-up, and replay to the un-captured result."""
    rng = np.random.default_rng(23)
    n, d = 20000, 5
    ref = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.02, axis=0).astype(np.float32)
    Lat = phl.Lattice(torch.from_numpy(ref).cuda())
    if layout == "odd_width":
        L = 231
        Lat.reserve(L)
        x = torch.rand((n, L), device="cuda")
        out = torch.empty_like(x)
    else:
        L = 160
        Lat.reserve(L, strided_io=True)
        x = torch.rand((n, L + 3), device="cuda")[:, 1:L + 1]
        out = torch.empty((n, L + 2), device="cuda")[:, 1:L + 1]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Lat.filter(x, out=out)
    g.replay()
    torch.cuda.synchronize()
    want = phl.Lattice(torch.from_numpy(ref).cuda()).filter(x.contiguous())
    assert float((out.contiguous() - want).abs().max()) <= 1e-6 * float(want.abs().max())

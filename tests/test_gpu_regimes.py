"""GPU parity across feature regimes (SURVEY.md 8d "synthetic inputs"; hard part 1 of SURVEY section 7:
"few fat vertices" vs "many thin vertices"), each against the CPU oracle on a crop the oracle finishes in seconds:

* natural-image features in the reference notebooks' three scalings (Experiments/DenseCrf.ipynb:142-146,
  crf/lattice/lite/test_bilateral.ipynb cell 6) on the stored Tsukuba frame, upsampled -- M/n 0.01 ... 0.5, vertices
  fed by hundreds of chunks (workgroup reduction of long partial-row lists), 256-entry segments (wave-cooperative
  sums), chunk vertex counts from 10 to 300 in one image (per-workgroup slab width in the slice, chunk classes in the
  splat);
* an image that mixes flat, smooth and iid-noise regions, so that one launch holds chunks of every kind including
  those whose vertex rows cannot be staged at all (the slice's direct form);
* the 8(d) stress case (iid colours).

Exact mode must be bit-identical to the oracle (= the reference engine, tests/test_oracle_golden.py), the default
mode within 1e-5 of the largest magnitude and 1e-4 per element on probabilities (north_star's tolerance).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def phl():
    import phl as _phl

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _phl.load_library()
    return _phl


def scaled_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_err(a, b):
    scale = np.maximum(np.abs(b), 1e-3 * max(np.abs(b).max(), 1e-30))
    return float((np.abs(a - b) / scale).max())


def _check_against_oracle(phl, feat, L, seed, expect=None):
    from oracle import phl_oracle as po

    n = feat.shape[0] * feat.shape[1]
    ref = np.ascontiguousarray(feat.reshape(n, -1))
    rng = np.random.default_rng(seed)
    src = rng.random((n, L), dtype=np.float32)
    src /= src.sum(1, keepdims=True)                      # probabilities, as mean-field feeds them
    want = po.oracle_filter(src, ref)
    lat = phl.Lattice(torch.from_numpy(ref).cuda())
    s = torch.from_numpy(src).cuda()
    stats = lat.tile_stats(L)
    if expect is not None:
        expect(lat, stats)
    exact = lat.filter(s, exact=True).cpu().numpy()
    assert np.array_equal(exact.view(np.uint32), want.view(np.uint32)), "exact mode differs from the CPU path"
    got = lat.filter(s).cpu().numpy()
    assert scaled_err(got, want) <= 1e-5
    assert rel_err(got, want) <= 1e-4
    again = lat.filter(s).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32)), "default mode is not reproducible"
    # the fused '- U' (LatticeGaussian, gaussian_matrix.py:303) takes the same kernels
    sub = lat.filter(s, subtract_input=True).cpu().numpy()
    assert np.abs(sub - (want - src)).max() <= 1e-5 * np.abs(want).max()
    # the staged kernels and the gather kernels must agree as well (same data structures, different walk)
    plain = lat.filter(s, no_tiles=True).cpu().numpy()
    assert scaled_err(plain, want) <= 1e-5
    return lat, stats


@pytest.mark.parametrize("sigma_c,sigma_p", [(0.1, 0.1), (0.08, 0.03), (0.125, 0.01)])
def test_natural_image_feature_scalings(phl, sigma_c, sigma_p):
    import bench

    H, W = 576, 768                                       # the stored 288x384 frame, upsampled 2x
    feat = bench.tsukuba_features(H, W, sigma_c, sigma_p)

    def expect(lat, stats):
        assert stats["staged_splat"] == 1 and stats["staged_slice"] == 1
        # fat vertices: far fewer vertices than pixels, and chunks of very different weight in one image
        assert lat.M < 0.6 * H * W

    lat, stats = _check_against_oracle(phl, feat, 32, 7, expect)
    print(f"tsukuba {sigma_c}/{sigma_p}: M/n {lat.M / (H * W):.4f} tiles {stats}")


def _mixed_image(H, W, sigma_xy):
    """Left third flat (one colour: 256-entry segments, vertices fed by hundreds of chunks), middle third smooth,
    right third iid noise (every pixel in its own simplex: chunks with ~1500 local vertices)."""
    import bench

    feat = bench.synthetic_features(H, W, sigma_xy=sigma_xy)
    rng = np.random.default_rng(99)
    feat[:, : W // 3, 2:] = np.float32(3.3)
    feat[:, 2 * W // 3:, 2:] = rng.random((H, W - 2 * W // 3, 3), dtype=np.float32) / np.float32(0.1)
    return feat


@pytest.mark.parametrize("L", [16, 64, 256])
def test_flat_smooth_and_noisy_regions_in_one_image(phl, L):
    H, W = 256, 768
    feat = _mixed_image(H, W, 30.0)

    def expect(lat, stats):
        # one image holds chunks the 256-channel slab takes whole and chunks nothing can stage
        assert stats["max_local_vertices"] > 1000, stats
        # ... and still runs on the staged kernels: per-workgroup slab widths in the slice, chunk
        # classes in the splat -- no chunk forces the whole image onto the gather kernels
        assert stats["staged_splat"] == 1 and stats["staged_slice"] == 1, stats

    lat, stats = _check_against_oracle(phl, feat, L, 3, expect)
    print(f"mixed image L={L}: M/n {lat.M / (H * W):.3f} tiles {stats}")


def test_iid_stress_case(phl):
    import bench

    H, W = 192, 256
    feat = bench.synthetic_features(H, W, iid=True)
    lat, stats = _check_against_oracle(phl, feat, 32, 5)
    assert lat.M > 3 * H * W                               # M/n -> d+1


def test_long_lists_and_long_segments_are_exercised(phl):
    """A flat colour over many chunks at a coarse position scale: few vertices, each fed by > 24 chunks (the
    workgroup reduction) through 256-entry segments (the wave-cooperative sums).  Checked against fp64."""
    H, W, L = 512, 512, 64
    feat = np.empty((H, W, 5), np.float32)
    feat[..., 0] = (np.arange(W, dtype=np.float32) / 200.0)[None, :]
    feat[..., 1] = (np.arange(H, dtype=np.float32) / 200.0)[:, None]
    feat[..., 2:] = 1.25
    ref = torch.from_numpy(feat.reshape(-1, 5)).cuda()
    lat = phl.Lattice(ref)
    stats = lat.tile_stats(L)
    assert stats["staged_splat"] == 1
    assert stats["slots"] > 24 * lat.M, "vertices should be fed by many chunks"
    rng = np.random.default_rng(11)
    src = rng.standard_normal((H * W, L)).astype(np.float32)
    s = torch.from_numpy(src).cuda()
    v_fast = lat.to_first_touch(lat.splat(s)).cpu().numpy()
    v_exact = lat.to_first_touch(lat.splat(s, exact=True)).cpu().numpy()
    # fp64 truth from the replay table
    vid, w = lat.replay()
    truth = np.zeros((lat.M, L), np.float64)
    for r in range(vid.shape[1]):
        np.add.at(truth, vid[:, r], w[:, r, None].astype(np.float64) * src)
    scale = np.abs(truth).max()
    e_fast, e_exact = np.abs(v_fast - truth).max() / scale, np.abs(v_exact - truth).max() / scale
    assert e_fast <= 2e-6 and e_exact <= 2e-5, (e_fast, e_exact)   # tree-like partial sums beat the reference's running sum
    again = lat.to_first_touch(lat.splat(s)).cpu().numpy()
    assert np.array_equal(v_fast.view(np.uint32), again.view(np.uint32))


def test_randomised_mixtures_of_chunk_kinds(phl):
    """Random images made of flat, smooth and noisy patches, random d, channel counts and position scales: every
    combination of (chunk class, slab width incl. the direct form, cooperative / plain segments, short / long
    partial-row lists) must give the oracle's result -- exact mode bit for bit, default mode to fp32 rounding."""
    from oracle import phl_oracle as po

    rng = np.random.default_rng(20261005)
    for trial in range(10):
        d = int(rng.choice([2, 3, 5, 5, 8]))
        vd = int(rng.choice([4, 20, 64, 100, 256]))
        H, W = int(rng.choice([64, 96, 128])), int(rng.choice([96, 160, 224]))
        sxy = float(rng.choice([2.0, 6.0, 25.0, 120.0]))
        feat = np.empty((H, W, d), np.float32)
        feat[..., 0] = (np.arange(W, dtype=np.float32) / sxy)[None, :]
        feat[..., 1] = (np.arange(H, dtype=np.float32) / sxy)[:, None]
        for k in range(2, d):
            feat[..., k] = 1.0
        # patches: 0 flat, 1 smooth ramp, 2 iid noise
        for by in range(0, H, 32):
            for bx in range(0, W, 32):
                kind = int(rng.integers(0, 3))
                blk = feat[by:by + 32, bx:bx + 32, 2:]
                if kind == 1:
                    blk[...] = (np.linspace(0, 3, blk.shape[1], dtype=np.float32)[None, :, None] + float(rng.random()))
                elif kind == 2:
                    blk[...] = rng.random(blk.shape, dtype=np.float32) * float(rng.choice([2.0, 10.0]))
        ref = np.ascontiguousarray(feat.reshape(-1, d))
        src = rng.standard_normal((H * W, vd)).astype(np.float32)
        want = po.oracle_filter(src, ref)
        lat = phl.Lattice(torch.from_numpy(ref).cuda())
        s = torch.from_numpy(src).cuda()
        exact = lat.filter(s, exact=True).cpu().numpy()
        assert np.array_equal(exact.view(np.uint32), want.view(np.uint32)), (trial, d, vd, H, W, sxy)
        got = lat.filter(s).cpu().numpy()
        assert scaled_err(got, want) <= 1e-5, (trial, d, vd, H, W, sxy, lat.tile_stats(vd))
        assert scaled_err(lat.filter(s, no_tiles=True).cpu().numpy(), want) <= 1e-5


def test_both_chunk_kernels_build_the_same_filter():
    """k_chunk_masks (bit masks + popcounts, chunks with <= 256 local vertices) and k_chunk_group (radix passes, any
    chunk; PHL_CHUNK_MASKS=0 sends every chunk through it) differ only in the order of a chunk's local vertex list --
    what is summed, and in which order, does not depend on it: the filter's output must be the same bit for bit, in
    the default arithmetic as well.  Fresh processes: the switch is read once."""
    import hashlib
    import subprocess

    code = (
        "import os, sys, hashlib, numpy as np, torch\n"
        "root = sys.argv[1]\n"
        "sys.path[:0] = [os.path.join(root, 'depth-estimation_amd'), root, os.path.join(root, 'tests')]\n"
        "import phl\n"
        "from test_gpu_regimes import _mixed_image\n"
        "feat = _mixed_image(256, 768, 30.0)\n"
        "ref = torch.from_numpy(np.ascontiguousarray(feat.reshape(-1, 5))).cuda()\n"
        "rng = np.random.default_rng(4)\n"
        "src = torch.from_numpy(rng.standard_normal((ref.shape[0], 64)).astype(np.float32)).cuda()\n"
        "L = phl.Lattice(ref)\n"
        "st = L.tile_stats(64)\n"
        "out = L.filter(src).cpu().numpy()\n"
        "print('RESULT', st['max_local_vertices'], st['slots'], hashlib.sha256(out.tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for masks in ("1", "0"):
        env = dict(os.environ, PHL_CHUNK_MASKS=masks)
        r = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
        res.append(line[1:])
    assert int(res[0][0]) > 384, "the image should hold chunks of all three kinds (<= 256, <= 384, more)"
    assert res[0] == res[1], res

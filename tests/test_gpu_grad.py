"""Fused backward of the lattice filter (phl_filter_grad, include/phl.h) against the reference's own formulation
-- one filter of the 2L(1+d)-channel operand [g, g(x)ref, src, src(x)ref] followed by the contraction of
crf/gaussian_matrix.py:450-463 -- which tests/golden/grad_*.npz pin to the reference's autograd output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _composition(lat, src, ref, g):
    from crf.gaussian_matrix import _ref_gradient, _wide_operand

    wall = lat.filter(_wide_operand(src, ref, g))
    return wall[:, :src.shape[1]], _ref_gradient(src, ref, g, wall)


def _features(kind, n, d, rng):
    if kind == "smooth":          # neighbouring pixels share vertices: the staged kernels' home ground
        return np.cumsum(rng.random((n, d), dtype=np.float32) * 0.05, axis=0).astype(np.float32)
    if kind == "image":
        side = int(np.sqrt(n))
        n = side * side
        yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
        cols = [xx.ravel() / 4, yy.ravel() / 4] + [np.sin(xx.ravel() / (7 + 3 * k)) * 2 + rng.random(n, dtype=np.float32) * 0.2 for k in range(d - 2)]
        return np.stack(cols[:d], 1).astype(np.float32)
    return (rng.random((n, d), dtype=np.float32) * 6).astype(np.float32)        # "random": little sharing, heavy chunks


@pytest.mark.parametrize("kind,n,d,L", [("smooth", 6000, 5, 32), ("smooth", 777, 3, 4), ("image", 64 * 64, 5, 64),
                                        ("image", 100 * 100, 3, 20), ("image", 48 * 48, 2, 256), ("random", 5000, 5, 16),
                                        ("smooth", 3000, 7, 8), ("image", 96 * 96, 5, 100)])
def test_fused_gradient_equals_the_wide_filter(kind, n, d, L):
    import phl

    rng = np.random.default_rng(n + d + L)
    f = _features(kind, n, d, rng)
    n = f.shape[0]
    ref = torch.from_numpy(f).cuda()
    src = torch.from_numpy(rng.random((n, L), dtype=np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32)).cuda()
    lat = phl.Lattice(ref)
    want_src, want_ref = _composition(lat, src, ref, g)
    try:
        got_src, got_ref = lat.filter_grad(src, g, ref)
    except phl.PhlError as e:
        assert e.status == 7 and kind == "random", e        # only the no-sharing case may be declined
        return
    assert float((got_src - want_src).abs().max()) <= 1e-5 * float(want_src.abs().max())
    scale = float(want_ref.abs().max())
    assert float((got_ref - want_ref).abs().max()) <= 2e-4 * scale, (float((got_ref - want_ref).abs().max()), scale)
    again = lat.filter_grad(src, g, ref, need_src=False)
    assert again[0] is None and torch.equal(again[1], got_ref), "not reproducible"


def test_autograd_through_the_mirrored_api_takes_the_fused_path(monkeypatch):
    """LatticeFilter.backward (the mirror of gaussian_matrix.py:435-468) with and without the fused kernels."""
    import crf.gaussian_matrix as gm

    rng = np.random.default_rng(5)
    n, d, L = 4000, 5, 16
    f = np.cumsum(rng.random((n, d), dtype=np.float32) * 0.05, axis=0).astype(np.float32)
    src = rng.random((n, L), dtype=np.float32)
    gout = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32)).cuda()

    def grads():
        ref = torch.from_numpy(f).cuda().requires_grad_(True)
        s = torch.from_numpy(src).cuda().requires_grad_(True)
        gm.LatticeFilter.apply(s, ref).backward(gout)
        return s.grad, ref.grad

    calls = []
    real = gm._fused_grad
    monkeypatch.setattr(gm, "_fused_grad", lambda *a: calls.append(1) or real(*a))
    gs_f, gr_f = grads()
    assert calls, "backward did not try the fused path"
    monkeypatch.setattr(gm, "_fused_grad", lambda *a: None)
    gs_c, gr_c = grads()
    assert float((gs_f - gs_c).abs().max()) <= 1e-5 * float(gs_c.abs().max())
    assert float((gr_f - gr_c).abs().max()) <= 2e-4 * float(gr_c.abs().max())


def test_full_size_timing_c2():
    """BASELINE configs[1] size (1390x1110x256, d=5): the fused backward runs where the reference's formulation
    would need 19 GB for its operand and 19 GB for the filtered result; prints the time for DESIGN.md."""
    import os
    import sys

    import phl

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    H, W, L, _ = bench.WORKLOADS["c2"]
    feat = bench.synthetic_features(H, W)
    dev = torch.device("cuda")
    ref = torch.from_numpy(feat.reshape(-1, 5)).to(dev)
    src = bench.synthetic_values(torch, H, W, L, 0, dev)
    g = torch.randn(src.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    lat = phl.Lattice(ref)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    for _ in range(2):
        gs, gr = lat.filter_grad(src, g, ref)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gs, gr = lat.filter_grad(src, g, ref)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    unit = src.numel() * 4
    lib_bytes = lat.device_bytes
    print(f"[measured] fused backward c2 {W}x{H}x{L}: {ms:.2f} ms; torch-side extra {(torch.cuda.max_memory_allocated() - base) / unit:.2f} x [n,L]; "
          f"lattice + workspaces {lib_bytes / unit:.2f} x [n,L]")
    assert torch.isfinite(gr).all() and torch.isfinite(gs).all()
    # a crop of it against the wide-filter formulation
    h, w = 160, 256
    idx = (torch.arange(h, device=dev)[:, None] * W + torch.arange(w, device=dev)[None, :]).flatten()
    r2, s2, g2 = ref[idx].contiguous(), src[idx].contiguous(), g[idx].contiguous()
    l2 = phl.Lattice(r2)
    want_src, want_ref = _composition(l2, s2, r2, g2)
    got_src, got_ref = l2.filter_grad(s2, g2, r2)
    assert float((got_ref - want_ref).abs().max()) <= 2e-4 * float(want_ref.abs().max())
    assert float((got_src - want_src).abs().max()) <= 1e-5 * float(want_src.abs().max())
    # (no wall-clock assertion in the parity suite: bench.py reports the time as `backward_c2`)


def test_batched_backward_takes_the_fused_path_item_by_item(monkeypatch):
    """BatchedLatticeFilter.backward (gaussian_matrix.py:396-421): [bs, n, L] / [bs, n, d] views of NCHW tensors."""
    import crf.gaussian_matrix as gm

    rng = np.random.default_rng(8)
    bs, h, w, d, L = 2, 40, 48, 5, 8
    n = h * w
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    guide = np.stack([np.stack([xx / 4, yy / 4] + [np.sin(xx / (5 + k + b)) + rng.random((h, w), dtype=np.float32) * 0.1 for k in range(d - 2)])
                      for b in range(bs)]).astype(np.float32)               # [bs, d, h, w]
    srcs_nchw = torch.from_numpy(rng.random((bs, L, h, w), dtype=np.float32)).cuda()
    gout = torch.from_numpy(rng.standard_normal((bs, n, L)).astype(np.float32)).cuda()

    def grads():
        refs = torch.from_numpy(guide).cuda().reshape(bs, d, n).permute(0, 2, 1).requires_grad_(True)
        srcs = srcs_nchw.reshape(bs, L, n).permute(0, 2, 1).detach().requires_grad_(True)
        gm.BatchedLatticeFilter.apply(srcs, refs).backward(gout)
        return srcs.grad, refs.grad

    import phl

    phl.clear_cache()
    built = []
    real = phl.Lattice

    class Counting(real):
        def __init__(self, *a, **k):
            built.append(1)
            super().__init__(*a, **k)

    monkeypatch.setattr(phl, "Lattice", Counting)
    gs_f, gr_f = grads()
    assert len(built) == bs, f"the backward pass must meet the lattices the forward pass cached ({len(built)} builds for {bs} items)"
    monkeypatch.setattr(gm, "_fused_grad", lambda *a: None)
    gs_c, gr_c = grads()
    assert float((gs_f - gs_c).abs().max()) <= 1e-5 * float(gs_c.abs().max())
    assert float((gr_f - gr_c).abs().max()) <= 2e-4 * float(gr_c.abs().max())


def test_one_phase_wide_splat_equals_the_multi_phase_form_bitwise(monkeypatch):
    """k_splat_wide (every LDS row read feeds all d+1 accumulators) against k_splat_tiled's nsets > 1 mode (the sum
    phase repeated per set): same products, same summation order -> the same gradients bit for bit."""
    import phl

    rng = np.random.default_rng(21)
    side, d, L = 112, 5, 128
    f = _features("image", side * side, d, rng)
    n = f.shape[0]
    ref = torch.from_numpy(f).cuda()
    src = torch.from_numpy(rng.random((n, L), dtype=np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32)).cuda()
    lat = phl.Lattice(ref)
    monkeypatch.setenv("PHL_WIDE_ONE_PHASE", "1")
    a_src, a_ref = lat.filter_grad(src, g, ref)
    monkeypatch.setenv("PHL_WIDE_ONE_PHASE", "0")
    b_src, b_ref = lat.filter_grad(src, g, ref)
    assert torch.equal(a_src, b_src) and torch.equal(a_ref, b_ref)


def test_channel_groups_under_a_small_workspace_budget(monkeypatch):
    """phl_filter_grad runs in channel groups when (2M + S_multi)(1+d)L floats exceed its workspace budget
    (PHL_GRAD_WS_MB): the contraction is a sum over channels, so the groups add up to the same gradients (fp32 order of
    the per-pixel channel sum differs between groupings: compared at the tolerance of the other tests), and the source
    gradient -- channel-wise -- is bit-identical."""
    import phl

    rng = np.random.default_rng(33)
    side, d, L = 96, 5, 192
    f = _features("image", side * side, d, rng)
    n = f.shape[0]
    ref = torch.from_numpy(f).cuda()
    src = torch.from_numpy(rng.random((n, L), dtype=np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32)).cuda()
    lat = phl.Lattice(ref)
    a_src, a_ref = lat.filter_grad(src, g, ref)
    st = lat.tile_stats(L)
    rows = 2 * lat.M + st["multi_chunk_slots"]
    per_ch_mb = rows * (d + 1) * 4 / 2 ** 20
    monkeypatch.setenv("PHL_GRAD_WS_MB", str(max(1, int(per_ch_mb * 70))))      # room for one 64-channel group
    before = lat.device_bytes
    b_src, b_ref = lat.filter_grad(src, g, ref)
    assert torch.equal(a_src, b_src)
    assert float((a_ref - b_ref).abs().max()) <= 2e-5 * float(a_ref.abs().max())
    monkeypatch.setenv("PHL_GRAD_WS_MB", "1")                                   # not even four channels: declined, not allocated
    if per_ch_mb * 4 > 1:
        with pytest.raises(phl.PhlError) as ei:
            lat.filter_grad(src, g, ref)
        assert ei.value.status == 7
    assert lat.device_bytes <= before


@pytest.mark.parametrize("L", [231, 6, 1])
def test_label_counts_off_the_four_channel_grid_take_the_fused_path_padded(L, monkeypatch):
    """The reference's own label counts (w // 6 = 231 at 1390 columns, crf/depth.py:40) are not multiples of 4:
    LatticeFilter.backward pads src and g with zero channels (exact zeros in every term of gaussian_matrix.py:450-463's
    contraction) instead of materialising the 2L(1+d)-channel operand."""
    import crf.gaussian_matrix as gm
    import phl

    rng = np.random.default_rng(100 + L)
    n, d = 60 * 50, 5
    f = _features("image", n, d, rng)
    n = f.shape[0]
    src = rng.random((n, L), dtype=np.float32)
    gout = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32)).cuda()
    widths = []
    real = phl.Lattice.filter_grad
    monkeypatch.setattr(phl.Lattice, "filter_grad", lambda self, s, g, r, **k: widths.append(s.shape[1]) or real(self, s, g, r, **k))

    def grads():
        ref = torch.from_numpy(f).cuda().requires_grad_(True)
        s = torch.from_numpy(src).cuda().requires_grad_(True)
        gm.LatticeFilter.apply(s, ref).backward(gout)
        return s.grad, ref.grad

    gs_f, gr_f = grads()
    assert widths == [(L + 3) // 4 * 4], widths
    assert gs_f.shape == (n, L) and gs_f.is_contiguous()
    monkeypatch.setattr(gm, "_fused_grad", lambda *a: None)       # the reference's formulation through the same lattice
    gs_c, gr_c = grads()
    assert float((gs_f - gs_c).abs().max()) <= 1e-5 * float(gs_c.abs().max())
    assert float((gr_f - gr_c).abs().max()) <= 2e-4 * float(gr_c.abs().max())

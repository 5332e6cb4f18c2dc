"""CPU: the import surface Experiments/DenseCrf.ipynb needs (cells 1-2) exists, and the caller-side
helpers behave (readers round-trip files, cost volume shape / argmin)."""
import numpy as np


def test_densecrf_notebook_imports():
    from crf.gaussian_matrix import GuidedAdjacency, LatticeGaussian, RbfLaplacian  # noqa: F401  (cell 1)
    from crf.utils import read_image, read_pfm, read_pgm  # noqa: F401                (cell 2)
    from crf.features import Vgg16features  # noqa: F401
    import crf.crf as legacy
    import crf.depth as depth

    for name in ("mean_field_infer", "charbonneir", "compatibility_matrix", "gaussian_weights_u"):
        assert hasattr(legacy, name)
    for name in ("disparity_badness", "disparity_estimate", "AD", "SD"):
        assert hasattr(depth, name)


def test_star_imports_give_the_names_the_other_notebooks_use():
    """trainableDenseCRF / Spectral_clustering / benchmarking.ipynb do ``from crf.<module> import *`` and then use
    module-level names of the reference modules (np, F, time, sp, identity_op, ...)."""
    import crf.crf as legacy
    import crf.depth as depth
    import crf.features as features
    import crf.gaussian_matrix as gm

    for name in ("np", "F", "time", "LatticeGaussian", "LatticeFilter", "RbfLaplacian", "RbfLaplacianC", "GuidedAdjacency",
                 "BatchedAdjacency", "latticefilter"):
        assert hasattr(gm, name), name
    for name in ("sp", "scipy", "np", "identity_op", "diag_op", "laplacian_op", "convolve_op", "normalized", "centroids"):
        assert hasattr(depth, name), name
    assert hasattr(features, "F") and hasattr(legacy, "mean_field_infer")
    op = depth.identity_op((4, 5)) + 2.0 * depth.laplacian_op((4, 5))       # composes like in Spectral_clustering.ipynb
    v = np.arange(20, dtype=float)
    assert np.allclose(op @ v, v + 2.0 * depth.laplacian(v.reshape(4, 5)).reshape(-1))
    assert depth.sp.sparse.linalg.LinearOperator is not None
    img = np.random.default_rng(0).random((6, 7, 3))
    assert abs(depth.normalized(img).mean()) < 1e-6


def test_readers_round_trip(tmp_path):
    from PIL import Image

    from crf.utils import read_image, read_pfm, read_pgm

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    Image.fromarray(img).save(tmp_path / "a.png")
    assert np.allclose(read_image(tmp_path / "a.png"), img / 255.0)
    g = rng.integers(0, 256, (5, 6), dtype=np.uint8)
    (tmp_path / "g.pgm").write_bytes(b"P5\n# comment\n6 5\n255\n" + g.tobytes())
    assert np.array_equal(read_pgm(tmp_path / "g.pgm"), g)
    d = rng.random((4, 3)).astype("<f4")
    (tmp_path / "d.pfm").write_bytes(b"Pf\n3 4\n-1.0\n" + d.tobytes())
    assert np.allclose(read_pfm(tmp_path / "d.pfm"), np.flip(d, axis=0))


def test_cost_volume_finds_a_known_shift():
    from crf.depth import AD, disparity_badness, disparity_estimate

    rng = np.random.default_rng(1)
    right = rng.random((20, 96, 3))
    left = np.roll(right, 5, axis=1)               # true disparity 5 everywhere (away from the wrap)
    E = disparity_badness(left, right, 9, AD)
    assert E.shape == (20, 96, 16)
    est = disparity_estimate(left, right)
    assert (est[:, 30:90] == 5).mean() > 0.99


def test_vgg_features_offline_placeholder():
    import warnings

    from crf.features import Vgg16features

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = Vgg16features()
    feats = net.get_all_features(np.random.default_rng(2).random((24, 32, 3)))
    assert [f.shape[-1] for f in feats] == [64, 128, 256, 512] and all(f.shape[:2] == (24, 32) for f in feats)


def test_stereo_heads_run_on_cpu_with_guided_w():
    """crf.mb_stereo_crf nn.Modules (API surface; guided-filter W, plain torch)."""
    import torch

    from crf.mb_stereo_crf import CRFdepthRefiner, CRFdepthUpsampler, CRFwUncertainty, logits2average_depth

    torch.manual_seed(0)
    logits, rgb, feats = torch.randn(1, 5, 40, 40), torch.rand(1, 3, 40, 40), torch.randn(1, 8, 40, 40)
    assert CRFdepthRefiner(d_in=8, d_guide=6, r=4, niters=1)((logits, rgb, feats)).shape == (1, 1, 40, 40)
    depth, conf = CRFwUncertainty(d_in=8, d_guide=6, r=4, niters=1)((logits, rgb, feats))
    assert depth.shape == conf.shape == (1, 1, 40, 40) and bool((conf > 0).all())
    up = CRFdepthUpsampler(r=4, niters=1)((torch.rand(1, 1, 10, 10) * 5, rgb, None))
    assert up.shape == (1, 1, 40, 40) and bool(torch.isfinite(up).all())
    assert torch.allclose(logits2average_depth(torch.zeros(1, 4, 2, 2)), torch.full((1, 1, 2, 2), 1.5))

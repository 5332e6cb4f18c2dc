"""GPU: phl_cost_volume (csrc/phl_costvol.hip) against the reference's own outputs (goldens) and against
the float64 numpy oracle on shapes the goldens do not cover.  fp32 on the device vs float64 in the
reference: asserted at 2e-5 of the volume's largest value (north-star tolerance is 1e-4)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def scaled_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


def test_goldens_from_the_reference(golden_dir):
    import phl

    files = sorted(glob.glob(os.path.join(golden_dir, "costvol_*.npz")))
    assert len(files) >= 5
    for f in files:
        g = np.load(f)
        h, w, L = g["out"].shape
        E = phl.cost_volume(g["img1"], g["img2"], window_size=int(g["window"]), criterion=str(g["criterion"]))
        assert E.shape == (h * w, L) and E.dtype == torch.float32 and E.is_cuda
        assert scaled_err(E.cpu().numpy().reshape(h, w, L).astype(np.float64), g["out"]) <= TOL, f


@pytest.mark.parametrize("h,w,c,ws,L,crit", [(37, 53, 3, 9, None, "AD"), (8, 16, 3, 1, 5, "SD"), (50, 200, 3, 9, 70, "AD"),
                                             (5, 7, 2, 7, 20, "AD"), (64, 48, 1, 11, 33, "nprod"), (1, 40, 3, 3, 6, "SD"),
                                             (40, 1, 3, 5, 2, "AD"), (24, 100, 4, 13, 64, "AD")])
def test_random_shapes_against_oracle(h, w, c, ws, L, crit):
    """ragged tiles, disparities beyond the image width, windows larger than the image, 1..4 channels,
    disparity counts off the 32-wide block."""
    import phl
    from oracle import costvol_oracle as co

    rng = np.random.default_rng(h * 1000 + w)
    a, b = rng.random((h, w, c)), rng.random((h, w, c))
    want = co.disparity_badness(a, b, ws, crit, max_disp=L)
    got = phl.cost_volume(torch.from_numpy(a), torch.from_numpy(b).cuda(), max_disp=L, window_size=ws, criterion=crit)
    assert got.shape == (h * w, want.shape[2])
    assert scaled_err(got.cpu().numpy().reshape(want.shape).astype(np.float64), want) <= TOL


def test_errors_and_device_entry_point():
    import phl
    from crf import depth

    rng = np.random.default_rng(1)
    a, b = rng.random((20, 60, 3)), rng.random((20, 60, 3))
    with pytest.raises(phl.PhlError) as e:
        phl.cost_volume(a, b, window_size=4)
    assert e.value.status == 7                      # PHL_ERR_UNSUPPORTED: even window
    with pytest.raises(phl.PhlError):
        phl.cost_volume(rng.random((4, 8, 5)), rng.random((4, 8, 5)))      # 5 channels
    with pytest.raises(ValueError):
        phl.cost_volume(a, b[:, :-1])
    assert phl.cost_volume(a, b, max_disp=0).shape == (1200, 0)
    E0 = depth.disparity_energy_device(a, b)        # window 9, AD, L = w // 6: the notebook's call
    want = depth.disparity_badness(a, b, 9, depth.AD)
    assert scaled_err(E0.cpu().numpy().reshape(want.shape).astype(np.float64), want) <= TOL
    # winner-takes-all disparity agrees wherever the float64 margin exceeds the fp32 error
    srt = np.sort(want, -1)
    clear = (srt[..., 1] - srt[..., 0]) > 1e-4 * want.max()
    assert np.array_equal(E0.argmin(1).cpu().numpy().reshape(20, 60)[clear], want.argmin(-1)[clear])
    # strided output: a column block of a wider E_0 buffer
    big = torch.zeros((1200, 32), device="cuda")
    phl.cost_volume(a, b, out=big[:, 8:18])
    assert torch.equal(big[:, 8:18], E0) and float(big[:, :8].abs().sum()) == 0.0 and float(big[:, 18:].abs().sum()) == 0.0

"""Unary stereo costs the reference notebooks import from ``crf.depth`` (caller side of the hot
path: they only produce the INPUT E_0 of the mean-field inference).  Mirrors
crf/depth.py:24-53 (SD, AD, nprod, disparity_estimate, disparity_badness); numpy/scipy only."""
import numpy as np
from scipy import ndimage


def SD(imga, imgb):
    return (imga - imgb) ** 2


def AD(imga, imgb):
    return np.abs(imga - imgb)


def nprod(imga, imgb):
    return -1 * imga * imgb


def disparity_badness(img1, img2, window_size=9, criterion=AD):
    """Cost volume [h, w, w // 6]: per disparity the criterion between the left image and the
    right image shifted by that disparity, summed over colour, box-aggregated over
    window_size x window_size (reflecting borders, like scipy's default convolve)."""
    max_disp = img1.shape[1] // 6
    h, w, _ = img1.shape
    padded = np.pad(img2, ((0, 0), (max_disp, 0), (0, 0)), mode="constant")
    cost = np.empty((h, w, max_disp))
    for disp in range(max_disp):
        cost[:, :, disp] = criterion(img1, padded[:, max_disp - disp:max_disp - disp + w]).sum(2)
    return ndimage.convolve(cost, np.ones((window_size, window_size, 1)))


def disparity_estimate(img1, img2, window_size=9, criterion=AD):
    """Winner-takes-all disparity of the window sweep."""
    return np.argmin(disparity_badness(img1, img2, window_size, criterion), axis=-1)


def disparity_energy_device(img1, img2, window_size=9, criterion=AD, max_disp=None):
    """``disparity_badness`` computed on the GPU (csrc/phl_costvol.hip) and left there as the
    E_0 [h*w, L] fp32 tensor ``mean_field_infer`` consumes (DenseCrf.ipynb cell 7 does
    ``torch.from_numpy(disp_energy.reshape(-1, L)).float().to(device)`` after the CPU sweep)."""
    import phl

    return phl.cost_volume(img1, img2, max_disp=max_disp, window_size=window_size, criterion=criterion)

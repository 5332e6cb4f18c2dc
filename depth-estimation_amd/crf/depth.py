"""Unary stereo costs the reference notebooks import from ``crf.depth`` (caller side of the hot
path: they only produce the INPUT E_0 of the mean-field inference).  Mirrors
crf/depth.py:24-53 (SD, AD, nprod, disparity_estimate, disparity_badness); numpy/scipy only."""
import numpy as np
import scipy
import scipy as sp            # the notebooks reach scipy through ``from crf.depth import *`` (Spectral_clustering.ipynb: sp.sparse.linalg)
import scipy.sparse.linalg
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


# ---- small numpy / scipy helpers of the same module (crf/depth.py:10-22, :102-147), used by the notebooks
# around the spectral-clustering experiments; nothing here touches the GPU path ---------------------------
def normalized(img, window_shape=None):
    """Zero-mean / unit-variance image, globally or over a sliding window (crf/depth.py:10-22)."""
    if window_shape is None:
        mean = lambda a: a.mean(axis=(0, 1))
    else:
        box = np.ones(window_shape) / (window_shape[0] * window_shape[1])
        box = box[..., None] if img.ndim == 3 else box
        mean = lambda a: ndimage.convolve(a, box)
    centred = img - mean(img)
    return centred / (np.sqrt(mean(centred ** 2)) + 1e-6)


def centroids(masks):
    """(mean i, mean j) of every mask of an [k, h, w] stack (crf/depth.py:102-105)."""
    ii, jj = np.mgrid[0:masks.shape[1], 0:masks.shape[2]]
    return np.array([(ii * masks).mean((1, 2)), (jj * masks).mean((1, 2))])


def laplacian(img):
    """5-point Laplacian with zero borders (crf/depth.py:113-116)."""
    return ndimage.convolve(img, np.array([[0, -1, 0], [-1, 4, -1], [0, -1, 0]]), mode="constant")


def convolve_op(filter, img_shape):
    """scipy LinearOperator applying a 2-D stencil to a flattened image (crf/depth.py:118-124)."""
    n = img_shape[0] * img_shape[1]
    return scipy.sparse.linalg.LinearOperator(
        (n, n), lambda v: ndimage.convolve(v.reshape(img_shape), filter, mode="constant").reshape(n))


def laplacian_op(img_shape):
    return convolve_op(np.array([[0, -1, 0], [-1, 4, -1], [0, -1, 0]]), img_shape)


def identity_op(img_shape):
    """Identity as a LinearOperator over flattened images (crf/depth.py:138-140; Spectral_clustering.ipynb adds
    1e-4 of it to the Laplacian before eigsh)."""
    n = img_shape[0] * img_shape[1]
    return scipy.sparse.linalg.LinearOperator((n, n), lambda v: v)


def diag_op(img):
    flat = img.reshape(-1)
    return scipy.sparse.linalg.LinearOperator((flat.size, flat.size), lambda v: flat * v)

"""CPU restatement of the reference's unary cost volume -- TEST INFRASTRUCTURE ONLY (imported by
tests/, tests/golden/generate.py and nothing else; the product path is csrc/phl_costvol.hip).

Follows crf/depth.py:36-53 (``disparity_badness``) and its criteria :24-29, float64 like the
reference.  Pinned bit for bit against the reference's own function by tests/golden/generate.py
(``costvol_*.npz``: the reference's outputs on a Tsukuba crop and on random images).
"""
import numpy as np
from scipy import ndimage

CRITERIA = {
    "SD": lambda a, b: (a - b) ** 2,          # crf/depth.py:24-25
    "AD": lambda a, b: np.abs(a - b),         # :26-27
    "nprod": lambda a, b: -1 * a * b,         # :28-29
}


def disparity_badness(img1, img2, window_size=9, criterion="AD", max_disp=None):
    """out[y, x, k] = box_{ws x ws, scipy 'reflect'}( sum_ch crit(img1[y, x], img2[y, x - k]) ), img2 = 0 left of the
    image; k = 0 .. max_disp-1, max_disp = w // 6 as the reference fixes it (:40) unless given."""
    crit = CRITERIA[criterion]
    h, w, _ = img1.shape
    L = w // 6 if max_disp is None else int(max_disp)
    padded = np.pad(img2, ((0, 0), (L, 0), (0, 0)), mode="constant")                 # :44
    out = np.zeros((h, w, L))                                                       # :46
    for i in range(L):                                                              # :47-50
        out[:, :, i] = crit(img1, padded[:, L - i:w + L - i]).sum(2)
    return ndimage.convolve(out, np.ones((window_size, window_size, 1)))            # :51-52

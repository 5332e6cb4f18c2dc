"""Guided-filter adjacency operators -- API surface only.

The reference's CRFasRNN defaults to a guided-filter W (crf/gaussian_matrix.py:161-287,
crf_module.py:91).  That is a different kernel from the lattice hot path (box sums over dense
NCHW tensors, already GPU-resident torch ops) and is OUT OF SCOPE for the HIP work
(SURVEY.md section 2.1).  These classes keep the constructor / call signatures so that code
written against the reference imports and runs; they are plain torch and their numerics are
"parity unpinned": the reference builds on the pip package ``guided_filter_pytorch`` (BoxFilter),
which is absent from this image, so no reference output exists to pin them against.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _box_sum(x, r):
    """Sum over a (2r+1)^2 window with zero padding, NCHW, via 2-D prefix sums."""
    c = F.pad(x, (r + 1, r, r + 1, r)).cumsum(2).cumsum(3)
    k = 2 * r + 1
    return c[:, :, k:, k:] - c[:, :, :-k, k:] - c[:, :, k:, :-k] + c[:, :, :-k, :-k]


class GuidedFilter(nn.Module):
    """y filtered by guide x with per-guide-channel (diagonal covariance) linear model,
    eps = softplus(omega) as in the reference's parametrisation (:161-232)."""

    def __init__(self, channels=1, r=20, eps=1e-8, gaussian=False):
        super().__init__()
        if gaussian:
            raise NotImplementedError("gaussian=True (learned-sigma box cascade) is outside the lattice scope")
        self.omega = nn.Parameter(torch.log(torch.expm1(torch.tensor(float(eps)))).expand(channels).clone())
        self._r = r
        self.gaussian = False

    def r(self):
        return self._r

    @property
    def eps(self):
        return F.softplus(self.omega)

    def _window(self):
        return self._r

    def get_coeffs(self, y, x):
        n, cx, h, w = x.shape
        cy = y.shape[1]
        r = self._window()
        N = _box_sum(torch.ones((1, 1, h, w), dtype=x.dtype, device=x.device), r)
        mean = lambda t: _box_sum(t, r) / N
        mx, my = mean(x), mean(y)
        cov = mean((y[:, :, None] * x[:, None]).reshape(n, cy * cx, h, w)).reshape(n, cy, cx, h, w) - my[:, :, None] * mx[:, None]
        var = mean(x * x) - mx * mx
        A = cov / (var[:, None] + self.eps.view(1, 1, -1, 1, 1))          # [n, cy, cx, h, w]
        b = my - (A * mx[:, None]).sum(2)
        mean_A = mean(A.reshape(n, cy * cx, h, w)).reshape(n, cy, cx, h, w)
        return mean_A, mean(b)

    def forward(self, y, x):
        mean_A, mean_b = self.get_coeffs(y, x)
        return (mean_A * x[:, None]).sum(2) + mean_b


class FastGuidedFilter(GuidedFilter):
    """Coefficients solved at 1/subsample_ratio resolution and upsampled (:234-253)."""

    def __init__(self, *args, subsample_ratio=2, mode="nearest", **kwargs):
        super().__init__(*args, **kwargs)
        self.subsample_ratio = subsample_ratio
        self.mode = mode

    def _window(self):
        return self._r // self.subsample_ratio

    def forward(self, y, x):
        s = self.subsample_ratio
        n, cx, h, w = x.shape
        cy = y.shape[1]
        lo = (h // s, w // s)
        A, b = self.get_coeffs(F.interpolate(y, size=lo, mode=self.mode), F.interpolate(x, size=lo, mode=self.mode))
        A = F.interpolate(A.reshape(n, cy * cx, *lo), size=(h, w), mode=self.mode).reshape(n, cy, cx, h, w)
        b = F.interpolate(b, size=(h, w), mode=self.mode)
        return (A * x[:, None]).sum(2) + b


class BatchedGuidedAdjacency(FastGuidedFilter):
    """W(src) = guided(src) * (2r+1)^2 / 2 - src  (:285-287)."""

    def forward(self, src_imgs, guide_imgs):
        return super().forward(src_imgs, guide_imgs) * 0.5 * (2 * self.r() + 1) ** 2 - src_imgs


class GuidedAdjacency(GuidedFilter):
    """Flat ``W @ U`` form used by Experiments/DenseCrf.ipynb cell 9: guide [1, C, H, W], U [n, L]."""

    def __init__(self, guide_img, r, eps):
        super().__init__(guide_img.shape[1], r, eps)
        self.guide_img = guide_img.float()

    def __matmul__(self, U):
        h, w = self.guide_img.shape[-2:]
        img = U.t().reshape(1, -1, h, w).float().to(self.guide_img.device)
        with torch.no_grad():
            out = self(img, self.guide_img) * 0.5 * (2 * self._r + 1) ** 2 - img
        return out[0].reshape(U.shape[1], -1).t().to(U.device)

"""Stereo refinement / upsampling heads built on CRFasRNN, mirroring the nn.Modules of the
reference's crf/mb_stereo_crf.py (:62-66 logits2average_depth, :68-102 CRFdepthRefiner and
CRFwUncertainty, :138-163 CRFdepthUpsampler).  API surface only: the reference instantiates them
with the guided-filter W (out of the lattice scope); ``lattice=True`` switches W to the
permutohedral BatchedAdjacency.  The trainer classes (:14-60, :105-136) and ``__main__`` need the
un-vendored ``oil`` package and Middlebury data and are not mirrored."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from crf.crf_module import CRFasRNN, charb


def logits2average_depth(logits, labels=None):
    """Expected label under softmax(logits): [bs, L, h, w] -> [bs, 1, h, w] (:62-66)."""
    probs = F.softmax(logits, dim=1)
    if labels is None:
        labels = torch.arange(probs.shape[1], dtype=torch.float32, device=probs.device)[None, :, None, None]
    return (probs * labels).sum(1, keepdim=True)


class _CoordConv(nn.Module):
    """3x3 convolution that also sees the normalised (i, j) pixel coordinates (stands in for the
    reference's oil ``conv2d(..., coords=True)``)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin + 2, cout, 3, padding=1)

    def forward(self, x):
        bs, _, h, w = x.shape
        ii = torch.linspace(-1, 1, h, device=x.device)[None, None, :, None].expand(bs, 1, h, w)
        jj = torch.linspace(-1, 1, w, device=x.device)[None, None, None, :].expand(bs, 1, h, w)
        return self.conv(torch.cat([x, ii, jj], dim=1))


class CRFdepthRefiner(nn.Module):
    def __init__(self, d_in=64, d_guide=16, r=15, niters=2, eps=1e-2, gamma=.05, lattice=False):
        super().__init__()
        self.CRF = CRFasRNN(charb(gamma), niters=niters, r=r, eps=eps, gchannels=d_guide, lattice=lattice)
        self.projection = nn.Conv2d(d_in, d_guide - 3, kernel_size=1)

    def _guide(self, imgrgb, features):
        return torch.cat((imgrgb, self.projection(features)), dim=1)

    def forward(self, inputs):
        logits, imgrgb, features = inputs
        return logits2average_depth(self.CRF(self._guide(imgrgb, features), logits))


class CRFwUncertainty(CRFdepthRefiner):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.uncertainty_net = nn.Sequential(_CoordConv(3, 16), nn.GroupNorm(4, 16), nn.ReLU(),
                                             _CoordConv(16, 16), nn.GroupNorm(4, 16), nn.ReLU(),
                                             _CoordConv(16, 1))          # log sigma, bs x 1 x h x w

    def forward(self, inputs):
        logits, imgrgb, features = inputs
        confidence = torch.exp(-self.uncertainty_net(imgrgb))
        out = self.CRF(self._guide(imgrgb, features), logits, confidence)
        return logits2average_depth(out), confidence


class CRFdepthUpsampler(nn.Module):
    def __init__(self, d_in=64, d_guide=3, r=15, niters=2, eps=1e-2, gamma=.05, lattice=False):
        super().__init__()
        self.CRF = CRFasRNN(charb(gamma), niters=niters, r=r, eps=eps, gchannels=d_guide, lattice=lattice)

    def forward(self, inputs):
        disp_lowres, img_highres, _ = inputs
        up = F.interpolate(disp_lowres, size=img_highres.shape[2:], mode="bilinear", align_corners=False)
        labels = torch.linspace(0, float(up.max()), 18, device=up.device)
        logits = -10 * self.CRF.Mu.get_energies_from_scalar(up, labels[None, :, None, None])
        confidence = (up > 1e-2).float()
        out = self.CRF(img_highres, logits, confidence=confidence, labels=labels)
        return logits2average_depth(out, labels[None, :, None, None])

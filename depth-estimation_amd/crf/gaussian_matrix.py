"""Adjacency ("W") operators of the dense CRF, mirroring the lattice half of the reference's
crf/gaussian_matrix.py.  Every operator funnels into ``latticefilter(src, ref)``, which here
is the MI355X HIP lattice (phl.filter) instead of the reference's CPU extension
(gaussian_matrix.py:15-16).  There is no CPU implementation behind it.

Mirrored names (reference line numbers in crf/gaussian_matrix.py):
    latticefilter :16      LatticeFilter :423-468      BatchedLatticeFilter :379-421
    batched_filter :370    LatticeGaussian :292-303    RbfLaplacian :305-319
    RbfLaplacianC :321-338 BatchedAdjacency :341-352
The guided-filter siblings (GuidedFilter, FastGuidedFilter, BatchedGuidedAdjacency,
GuidedAdjacency; :161-287) are a different, dense-torch kernel outside the lattice hot path;
they are provided in crf.guided as plain torch ops for API completeness.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

import time  # noqa: F401   (the notebooks pick np, F, time up through ``from crf.gaussian_matrix import *``)

import numpy as np  # noqa: F401
import torch.nn.functional as F  # noqa: F401

import phl

latticefilter = phl.filter  # same symbol the reference binds at import (:16)


def _shape_msg(a, b):
    return "Incompatible shapes {}, and {}".format(a, b)


def _ref_gradient(src, ref, g, wall):
    """Gradient of sum(g * filter(src, ref)) w.r.t. ref from ONE wide filter call.

    ``wall`` = filter([g, g(x)ref, src, src(x)ref], ref), i.e. 2L(1+d) channels (:450-463).
    With W_ij = exp(-|f_i - f_j|^2 / 2)-like symmetric weights the contraction is
        -2 * sum_L ( s f * Wg  -  s * W(g f)  +  g f * Ws  -  g * W(s f) ).
    Shapes: src,g [..., n, L]; ref [..., n, d]; returns [..., n, d].
    """
    L, d = src.shape[-1], ref.shape[-1]
    wg, wgf, ws, wsf = torch.split(wall, [L, L * d, L, L * d], dim=-1)
    f = ref.unsqueeze(-2)                                   # [..., n, 1, d]
    s_, g_ = src.unsqueeze(-1), g.unsqueeze(-1)             # [..., n, L, 1]
    wgf = wgf.reshape(wgf.shape[:-1] + (L, d))
    wsf = wsf.reshape(wsf.shape[:-1] + (L, d))
    inner = (s_ * f) * wg.unsqueeze(-1) - s_ * wgf + (g_ * f) * ws.unsqueeze(-1) - g_ * wsf
    return -2 * inner.sum(-2)


def _fused_shape_ok(src, ref, g):
    return (g.shape == src.shape and src.shape[-1] > 0 and ref.shape[-1] <= 7 and torch.cuda.is_available()
            and src.dtype == torch.float32 and ref.dtype == torch.float32
            and os.environ.get("PHL_FUSED_GRAD", "1") != "0")         # A/B switch: the reference's formulation


def _fused_grad(src, ref, g, need_src):
    """(grad_src | None, grad_ref) through phl_filter_grad: the products with ``ref`` are formed on the splat
    weights and the contraction happens inside the slice, so the 2L(1+d)-channel operand and result of :450-463
    (19 GB each at 1390x1110x256) never exist.  None when the fused path does not take the shape (then the caller
    filters the wide operand, as the reference does).  ``src`` / ``ref`` may be [n, .] (one image) or [bs, n, .]
    (a batch: items dealt over the devices and lattices of the batched forward pass).  Any label count: see the
    padding below."""
    if not _fused_shape_ok(src, ref, g) or src.dim() != ref.dim() or src.dim() not in (2, 3):
        return None
    # label counts off the kernels' 4-channel grid (the reference's own w // 6: 231, 341) run padded with channels
    # that are exactly zero in src AND g: every term of the contraction carries a factor s or g, the source gradient
    # of a padding channel is the filter of zeros -- the extra channels add exact zeros and are cut off again
    L = src.shape[-1]
    pad = (-L) % 4
    if pad:
        src, g = F.pad(src, (0, pad)), F.pad(g, (0, pad))
    try:
        if src.dim() == 3:
            gs, gr = phl.batched_filter_grad(src, ref, g, need_src=need_src)
        else:
            gs, gr = phl.lattice_for(ref.detach()).filter_grad(src, g, ref, need_src=need_src)
            gs, gr = (gs.to(src.device) if gs is not None else None), gr.to(ref.device)
    except phl.PhlError as e:
        if e.status != 7:          # PHL_ERR_UNSUPPORTED: shape outside the fused path
            raise
        return None
    if pad and gs is not None:
        gs = gs[..., :L].contiguous()
    return gs, gr


def _wide_operand(src, ref, g):
    L, d = src.shape[-1], ref.shape[-1]
    gf = (g.unsqueeze(-1) * ref.unsqueeze(-2)).reshape(g.shape[:-1] + (L * d,))
    sf = (src.unsqueeze(-1) * ref.unsqueeze(-2)).reshape(src.shape[:-1] + (L * d,))
    return torch.cat([g, gf, src, sf], dim=-1)


class LatticeFilter(Function):
    """autograd wrapper of one lattice filter (:423-468).  The operator is symmetric, so the
    gradient w.r.t. ``source`` is another filter of the incoming gradient (:445-446); the
    gradient w.r.t. ``reference`` costs one 2L(1+d)-channel filter (:450-463).

    Deviation, documented: the reference's source-only branch computes the right value and then
    raises UnboundLocalError at its timing print (:467); this one returns the value."""

    @staticmethod
    def forward(ctx, source, reference):
        assert source.shape[0] == reference.shape[0], _shape_msg(source.shape, reference.shape)
        ctx.save_for_backward(source, reference)
        return latticefilter(source, reference)

    @staticmethod
    def backward(ctx, grad_output):
        src, ref = ctx.saved_tensors
        need_src, need_ref = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_source = grad_reference = None
        with torch.no_grad():
            g = grad_output.contiguous()
            if need_ref:
                fused = _fused_grad(src, ref, g, need_src)
                if fused is not None:
                    grad_source, grad_reference = fused
                else:
                    wall = latticefilter(_wide_operand(src, ref, g), ref)
                    grad_reference = _ref_gradient(src, ref, g, wall)
                    if need_src:
                        grad_source = wall[..., :src.shape[-1]]
            elif need_src:
                grad_source = latticefilter(g, ref)
        return grad_source, grad_reference


def batched_filter(flat_srcs, flat_refs, num_threads=None):
    """Independent lattice per batch item (:370-377).  The reference forks a process pool of
    ``num_threads`` CPU workers per call, one image per worker; here every item is one cached lattice on a
    GPU, the items dealt round-robin over the visible GPUs (phl.batch_devices: all of them for CPU tensors, the
    tensors' own device otherwise), each GPU on its own stream, no collective.  ``num_threads`` is accepted for
    signature parity."""
    return phl.batched_filter(flat_srcs, flat_refs)


class BatchedLatticeFilter(Function):
    """Batched [bs, n, L] / [bs, n, d] counterpart of LatticeFilter (:379-421)."""

    @staticmethod
    def forward(ctx, flat_srcs, flat_refs, num_threads=None):
        assert flat_srcs.shape[:2] == flat_refs.shape[:2], _shape_msg(flat_srcs.shape, flat_refs.shape)
        ctx.save_for_backward(flat_srcs, flat_refs)
        ctx.num_threads = num_threads
        return batched_filter(flat_srcs, flat_refs, num_threads)

    @staticmethod
    def backward(ctx, grad_output):
        srcs, refs = ctx.saved_tensors
        need_src, need_ref = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_source = grad_reference = None
        with torch.no_grad():
            g = grad_output
            if need_ref:
                # item by item through the fused kernels (phl_filter_grad) where they take the shape, on the devices
                # and cached lattices of the forward pass
                fused = _fused_grad(srcs, refs, g, need_src)
                if fused is not None:
                    grad_source, grad_reference = fused
                else:
                    wall = batched_filter(_wide_operand(srcs, refs, g), refs, ctx.num_threads)
                    grad_reference = _ref_gradient(srcs, refs, g, wall)
                    if need_src:
                        grad_source = wall[..., :srcs.shape[-1]]
            elif need_src:
                grad_source = batched_filter(g, refs, ctx.num_threads)
        return grad_source, grad_reference, None


class LatticeGaussian(nn.Module):
    """W with ``W @ U = filter(U, ref) - U`` (:292-303): the un-normalised unit-sigma Gaussian
    affinity with the self term (approximately) removed."""

    def __init__(self, ref):
        super().__init__()
        self.ref = ref

    def __matmul__(self, U):
        return self(U)

    def forward(self, U):
        if not (torch.is_grad_enabled() and (U.requires_grad or self.ref.requires_grad)):
            # inference: "- U" fused into the slice epilogue (bit-identical to the subtraction)
            assert U.shape[0] == self.ref.shape[0], _shape_msg(U.shape, self.ref.shape)
            return phl.lattice_for(self.ref.detach()).filter(U, subtract_input=True)
        return LatticeFilter.apply(U, self.ref) - U


class RbfLaplacian(nn.Module):
    """D - W, or I - D^-1/2 W D^-1/2 when ``normalize`` (:305-319); W includes the self term
    here (plain filter), D = W 1."""

    def __init__(self, ref, normalize=True):
        super().__init__()
        self.ref = ref
        n = ref.shape[0]
        self.shape = (n, n)
        self.normalize = normalize
        self.D = LatticeFilter.apply(torch.ones((n, 1), dtype=ref.dtype, device=ref.device), ref)

    def __matmul__(self, U):
        if self.normalize:
            rs = self.D.sqrt()
            return U - LatticeFilter.apply(U / rs, self.ref) / rs
        return self.D * U - LatticeFilter.apply(U, self.ref)


class RbfLaplacianC(LatticeGaussian):
    """Laplacians of the self-term-free W of LatticeGaussian (:321-338):
    'sym' -> U - W(U/sqrt D)/sqrt D,  'right' -> U - W(U/D),  anything else -> D U - W U."""

    def __init__(self, ref, normalize="sym"):
        super().__init__(ref)
        n = ref.shape[0]
        self.shape = (n, n)
        self.normalize = normalize
        self.D = LatticeGaussian.forward(self, torch.ones((n, 1), dtype=ref.dtype, device=ref.device))

    def __matmul__(self, U):
        W = lambda X: LatticeGaussian.forward(self, X)
        if self.normalize == "sym":
            rs = self.D.sqrt()
            return U - W(U / rs) / rs
        if self.normalize == "right":
            return U - W(U / self.D)
        return self.D * U - W(U)


class BatchedAdjacency(nn.Module):
    """NCHW front end (:341-352): src [bs, L, H, W], guide [bs, d, H, W]; one lattice per image;
    returns filter(src) - src in NCHW.  The [n, L] / [n, d] views of NCHW tensors are channel-
    major; the C ABI takes their strides as they are (no host copy, one device transpose)."""

    def __init__(self, num_threads=8):
        super().__init__()
        self.num_threads = num_threads

    def forward(self, src_imgs, guide_imgs):
        bs, L, h, w = src_imgs.shape
        d = guide_imgs.shape[1]
        flat_srcs = src_imgs.reshape(bs, L, h * w).permute(0, 2, 1)
        flat_refs = guide_imgs.reshape(bs, d, h * w).permute(0, 2, 1)
        out = BatchedLatticeFilter.apply(flat_srcs, flat_refs, self.num_threads)
        return out.permute(0, 2, 1).reshape(src_imgs.shape) - src_imgs


# guided-filter family: API surface only, implemented in crf.guided
from crf.guided import (BatchedGuidedAdjacency, FastGuidedFilter, GuidedAdjacency,  # noqa: E402,F401
                        GuidedFilter)

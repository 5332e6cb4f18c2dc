"""Dense-CRF mean-field inference, mirroring the reference's crf/crf_module.py.

    Q <- softmax(-E0);  repeat niters:  E <- E0 + (W @ Q) @ Mu;  Q <- softmax(-E)

``W`` is any object with ``@`` -- for the lattice path a crf.gaussian_matrix.LatticeGaussian,
whose product is one splat -> blur -> slice on the MI355X lattice built ONCE for the fixed
reference features (the reference rebuilds it on every iteration).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from crf.gaussian_matrix import BatchedAdjacency, BatchedGuidedAdjacency  # noqa: F401  (crf_module.py:5)


def gaussian_weights_u(f):
    """Dense brute-force W = exp(-|fi-fj|^2) - I for tiny n (crf_module.py:17-20)."""
    sq = torch.cdist(f, f).pow(2)
    return torch.exp(-sq) - torch.eye(f.shape[0], device=f.device, dtype=f.dtype)


def gaussian_weights(f):
    """Symmetrically normalised version of gaussian_weights_u, minus I (crf_module.py:8-15)."""
    W = gaussian_weights_u(f)
    s = W.sum(1).rsqrt()
    return s[:, None] * W * s[None, :] - torch.eye(f.shape[0], device=f.device, dtype=f.dtype)


def lazy_W(f):
    """Row generator of the normalised dense affinity for an [h, w, c] numpy image (crf_module.py:22-30)."""
    def W(i, j):
        a = np.exp(-((f - f[i, j]) ** 2).sum(-1).reshape(-1))
        a = a / np.sqrt(a.sum() - 1)
        return a.reshape(f.shape[:2])
    return W


def charbonneir(a, b, gamma=.1):
    """Charbonnier label compatibility sqrt(gamma^2 + (a-b)^2) - gamma (crf_module.py:32-33)."""
    return torch.sqrt(gamma ** 2 + (a - b) ** 2) - gamma


def charbonneir2(a, b, gamma=3):
    return torch.sqrt(1 + ((a - b) / gamma) ** 2) - 1


def compatibility_matrix(compat, labels):
    """Mu[a, b] = compat(labels[a], labels[b]) (crf_module.py:38-39)."""
    return compat(labels[:, None], labels[None, :])


# ---- label counts the fused kernels cannot take as they are ------------------------------------------------------
# The reference takes max_disp = w // 6 (crf/depth.py:40): 231 labels at Middlebury's 1390 columns, 341 at 2048 -- rows
# that are not made of 16-byte pieces.  The device loop then runs on L rounded up (to a multiple of 4; to 256 when that is
# 32 labels away at most): the extra labels get an energy nothing else reaches, so their probability is exactly 0 in every
# iteration (exp underflows), their rows and columns of Mu are 0, W maps a zero column to a zero column -- the first L
# columns evolve exactly as without them.  One padded copy of E_0 on entry, one slice on exit.
_PAD_ENERGY = 1.0e30
_mu_pad_cache = {}


def _label_pad(L):
    if L % 4 == 0:
        return L
    return 256 if 224 < L < 256 else (L + 3) // 4 * 4


def _padded_mu(Mu, Lp, device):
    """Mu with zero rows / columns up to Lp, cached per (storage, version) like the kernels' own forms of it, plus the
    Potts-family structure of the ORIGINAL matrix (the padded one no longer shows it)."""
    import phl

    key = (Mu.data_ptr(), Mu._version, tuple(Mu.shape), tuple(Mu.stride()), Lp, str(device))
    hit = _mu_pad_cache.get(key)
    if hit is None:
        if len(_mu_pad_cache) > 8:
            _mu_pad_cache.clear()
        L = Mu.shape[0]
        mp = torch.zeros((Lp, Lp), dtype=torch.float32, device=device)
        mp[:L, :L] = Mu.detach().to(device, torch.float32)
        hit = _mu_pad_cache[key] = (mp, phl._mu_uniform(Mu.detach()) or False, Mu)
    return hit[0], hit[1]


def _pad_energies(E_0, Lp):
    n, L = E_0.shape
    E0p = torch.full((n, Lp), _PAD_ENERGY, dtype=torch.float32, device=E_0.device)
    E0p[:, :L] = E_0
    return E0p


def _fused_ok(E_0, Mu):
    return (E_0.is_cuda and E_0.dtype == torch.float32 and E_0.dim() == 2 and E_0.stride(1) == 1
            and not (torch.is_grad_enabled() and (E_0.requires_grad or Mu.requires_grad)))


def mean_field_step(E_0, W, Mu, Q, out=None):
    """One iteration ``softmax(-(E_0 + (W @ Q) @ Mu))`` (crf_module.py:51-52) on the fused device path.

    ``W`` is an operator (``@``) or a callable.  The raw-pointer kernels below know nothing of autograd, so
    the caller must have established that no gradient flows (``mean_field_infer`` does)."""
    import phl

    X = W(Q) if callable(W) and not hasattr(W, "__matmul__") else W @ Q
    if not (X.is_cuda and X.dtype == torch.float32 and X.stride(1) == 1):
        X = X.to(E_0.device, torch.float32).contiguous()
    return phl.compat_softmax(E_0, X, Mu, out=out)


def mean_field_infer(E_0, W, Mu, niters=10):
    """[E_0] n x L unaries, [W] n x n operator, [Mu] L x L compatibility -> Q n x L (crf_module.py:41-53).

    On the GPU inference path everything of an iteration outside the lattice filter -- the compatibility
    product, the add, the negation and the softmax -- is ONE fused HIP kernel (phl.compat_softmax: fp32 MFMA
    tiles of X @ Mu with the softmax as epilogue, so neither G nor E ever exists in HBM).  With autograd
    (E_0, Mu, or anything W carries -- e.g. a LatticeGaussian whose ``ref`` requires grad) or CPU tensors
    the plain, differentiable torch ops run."""
    if _fused_ok(E_0, Mu):
        import phl

        L = E_0.shape[1]
        Lp, uniform = _label_pad(L), None
        if Lp != L:                          # (see _label_pad: w // 6 labels are rarely a multiple of 4)
            E_0 = _pad_energies(E_0, Lp)
            Mu, uniform = _padded_mu(Mu, Lp, E_0.device)
        Q = phl.softmax_neg_add(E_0)
        fused = True
        for _ in range(niters):
            if fused:
                X = W @ Q
                if torch.is_grad_enabled() and X.requires_grad:
                    # W itself is being differentiated: the raw-pointer kernels would drop its graph (and
                    # overwrite a tensor LatticeFilter saved for backward) -> differentiable path from here on
                    fused = False
                    Q = F.softmax(-(E_0 + X @ Mu), dim=1)
                    continue
                if not (X.is_cuda and X.dtype == torch.float32 and X.stride(1) == 1):
                    X = X.to(E_0.device, torch.float32).contiguous()
                Q = phl.compat_softmax(E_0, X, Mu, out=Q, uniform=uniform)
            else:
                Q = F.softmax(-(E_0 + (W @ Q) @ Mu), dim=1)
        return Q if Lp == L else Q[:, :L].contiguous()
    if _staged_ok(E_0, W, Mu):
        return _mean_field_infer_staged(E_0, W, Mu, niters)
    Q = F.softmax(-E_0, dim=1)
    for _ in range(niters):
        Q = F.softmax(-(E_0 + (W @ Q) @ Mu), dim=1)
    return Q


def _staged_ok(E_0, W, Mu):
    """The notebook's own call shape (Experiments/DenseCrf.ipynb:142-152,173: CPU tensors, W = LatticeGaussian(ref) on
    the CPU, no autograd) -- everything the device loop needs can be staged once."""
    from crf.gaussian_matrix import LatticeGaussian

    if not (type(W) is LatticeGaussian and torch.is_tensor(W.ref) and torch.cuda.is_available()):
        return False
    no_grad = not (torch.is_grad_enabled() and (E_0.requires_grad or Mu.requires_grad or W.ref.requires_grad))
    return (no_grad and not E_0.is_cuda and E_0.dtype == torch.float32 and E_0.dim() == 2 and W.ref.dtype == torch.float32
            and Mu.dtype == torch.float32 and W.ref.dim() == 2 and W.ref.shape[0] == E_0.shape[0])


def _mean_field_infer_staged(E_0, W, Mu, niters):
    """CPU tensors in, CPU tensor out, the iterations on the device: E_0 crosses PCIe once (pinned pieces, phl.to_device),
    the lattice of ``W.ref`` is built once (cached per ``ref``), every iteration is one lattice filter with ``- Q`` fused
    and one fused compatibility + softmax kernel, and Q crosses PCIe once at the end -- instead of Q making the round
    trip inside every ``W @ Q`` with the compatibility product and the softmax on the host."""
    import phl

    dev = W.ref.device if W.ref.is_cuda else torch.device("cuda", torch.cuda.current_device())
    lat = phl.lattice_for(W.ref.detach())                 # CPU ref: built on the current device; GPU ref: where it lives
    E0d = phl.to_device(E_0.detach().contiguous(), dev)
    L = E0d.shape[1]
    Lp, uniform, Mu = _label_pad(L), None, Mu.detach()
    if Lp != L:
        E0d = _pad_energies(E0d, Lp)
        Mu, uniform = _padded_mu(Mu, Lp, dev)
    Q = phl.softmax_neg_add(E0d)
    X = torch.empty_like(Q) if niters > 0 else None
    for _ in range(niters):
        lat.filter(Q, subtract_input=True, out=X)
        Q = phl.compat_softmax(E0d, X, Mu, out=Q, uniform=uniform)   # (Mu^T is cached per Mu tensor, wherever it lives)
    return phl.to_host(Q if Lp == L else Q[:, :L].contiguous())


def potts(num_classes):
    """1x1 conv holding the Potts compatibility 1 - I (crf_module.py:55-64)."""
    conv = nn.Conv2d(num_classes, num_classes, kernel_size=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_((1 - torch.eye(num_classes))[..., None, None])
    return conv


def _compat_matrix(mu, L, labels, device):
    """[L, L] matrix M with ``mu(Q)[:, c] = sum_b M[b, c] Q[:, b]`` for the compatibility modules CRFasRNN is
    used with (``charb``, or a bias-free 1x1 conv such as ``potts``); None for anything else."""
    if isinstance(mu, charb):
        return mu.matrix(L, labels, device)
    if isinstance(mu, nn.Conv2d) and mu.kernel_size == (1, 1) and mu.bias is None and mu.groups == 1 and mu.weight.shape[:2] == (L, L):
        return mu.weight.detach()[:, :, 0, 0].t().to(device, torch.float32)      # conv: out[c] = sum_b w[c, b] q[b]
    return None


def _maybe_learnable(value, trainable):
    """A scalar hyper-parameter: nn.Parameter when it is to be trained, the plain number otherwise
    (the reference's ``nn.Parameter(torch.tensor(v)) if trainable else v`` idiom, crf_module.py:109-110)."""
    return nn.Parameter(torch.tensor(value)) if trainable else value


class charb(nn.Module):
    """Learnable Charbonnier compatibility applied as a 1x1 conv over label channels
    (crf_module.py:66-79): Mu(Q) = conv(Q, charbonneir(l_a, l_b, gamma)) * exp(s).
    Parameter names (``gamma``, ``s``) are the reference's, so its checkpoints load."""

    def __init__(self, gamma):
        super().__init__()
        self.register_parameter("gamma", nn.Parameter(torch.tensor(gamma)))
        self.register_parameter("s", nn.Parameter(torch.tensor(0.)))

    def _scale(self):
        return torch.exp(self.s)

    def forward(self, x, labels=None):
        if labels is None:          # the reference builds them with .cuda(); here they follow the input
            labels = torch.arange(x.shape[1], dtype=torch.float32, device=x.device)
        weight = compatibility_matrix(lambda p, q: charbonneir(q, p, self.gamma), labels)     # symmetric in (p, q)
        return F.conv2d(x, weight[..., None, None]) * self._scale()

    def matrix(self, L, labels=None, device=None):
        """The same compatibility as a dense [L, L] matrix for pixel-major data: forward(x)[:, c] = (x @ M)[:, c]."""
        if labels is None:
            labels = torch.arange(L, dtype=torch.float32, device=device)
        weight = compatibility_matrix(lambda p, q: charbonneir(q, p, self.gamma), labels.to(device))
        return (weight * self._scale()).detach().t().contiguous()

    def get_energies_from_scalar(self, x, labels):
        return charbonneir(labels, x, self.gamma * labels.max()) * self._scale()


def _mean_field_nchw(E0, message, niters):
    """NCHW mean field: ``message(Q)`` returns the pairwise energy; gives the energy of the last iteration."""
    E, Q = E0, F.softmax(-E0, dim=1)
    for _ in range(niters):
        E = E0 + message(Q)
        Q = F.softmax(-E, dim=1)
    return E


_nchw_streams = {}


def _mean_field_nchw_fused(E0, refs, M, niters):
    """The same iteration for a lattice W, without autograd, the way the device wants it: one transpose to
    pixel-major [n, L] per image on entry, then per iteration ONE lattice filter with the ``- Q`` fused
    (phl_filter) and ONE fused compatibility-product + softmax kernel (phl_compat_softmax), and one transpose
    back at the end -- Q, G and E never make extra passes over HBM.  The reference evaluates
    W(Mu(Q)); here (W Q) Mu: W acts on pixels, Mu on labels, so they commute (fp32 rounding differs).

    Batch items are independent (the reference hands them to a process pool, gaussian_matrix.py:370-377): they are
    dealt over phl.batch_devices (the tensors' own GPU, or all GPUs with PHL_BATCH_DEVICES=all / CPU inputs), two
    side streams per device, so that image b+1's lattice build and transposes run under image b's iterations; both
    transposes go through the library's LDS-tiled phl_copy2d."""
    import phl

    bs, L, h, w = E0.shape
    d = refs.shape[1]
    n = h * w
    if niters <= 0:
        return E0
    out = torch.empty_like(E0)
    home = E0.device
    devices = phl.batch_devices(E0)
    cur = torch.cuda.current_stream(home)
    used = []
    for b in range(bs):
        dev = devices[b % len(devices)]
        lane = (b // len(devices)) % 2
        st = _nchw_streams.get((dev, lane))
        if st is None:
            st = _nchw_streams[(dev, lane)] = torch.cuda.Stream(device=dev)
        st.wait_stream(cur)
        with torch.cuda.device(dev), torch.cuda.stream(st):
            e_b, r_b = E0[b], refs[b].detach()
            if dev != home:
                e_b, r_b = e_b.to(dev, non_blocking=True), r_b.to(dev, non_blocking=True)
            Lp = _label_pad(L)               # (label counts that are not a multiple of 4 run padded: see _label_pad)
            e0 = torch.empty((n, L), dtype=torch.float32, device=dev) if Lp == L else \
                torch.full((n, Lp), _PAD_ENERGY, dtype=torch.float32, device=dev)
            phl.copy2d(e0[:, :L], e_b.reshape(L, n).t())                  # [L, n] channel-major -> [n, L]
            lat = phl.lattice_for(r_b.reshape(d, n).t(), device=dev)      # strided [n, d] view, no copy
            Mb, uniform = (M if M.device == dev else M.to(dev)), None
            if Lp != L:
                Mb, uniform = _padded_mu(M, Lp, dev)
            Q = phl.softmax_neg_add(e0)
            for it in range(niters):
                X = lat.filter(Q, subtract_input=True)
                Q = phl.compat_softmax(e0, X, Mb, out=Q, logits=it == niters - 1, uniform=uniform)
            if dev == home:
                phl.copy2d(out[b].reshape(L, n).t(), Q[:, :L])
            else:
                back = torch.empty((L, n), dtype=torch.float32, device=dev)
                phl.copy2d(back.t(), Q[:, :L])
                out[b].reshape(L, n).copy_(back, non_blocking=True)
            for t in (e0, Q, e_b, r_b):
                t.record_stream(st)
        used.append(st)
    for st in used:
        cur.wait_stream(st)
    return out


class CRFasRNN(nn.Module):
    """Batched NCHW mean field (crf_module.py:81-104); returns logits -E of the LAST iteration.

    ``lattice=True`` selects the permutohedral W (BatchedAdjacency, the MI355X path); the default
    stays the reference's guided-filter W."""

    def __init__(self, mu_init, niters=5, r=20, eps=1e-5, notrain_mu=False, gaussian=False, gchannels=1, lattice=False):
        super().__init__()
        self.Mu, self.niters = mu_init, niters
        if notrain_mu:
            self.Mu.requires_grad_(False)
        self.W = BatchedAdjacency() if lattice else BatchedGuidedAdjacency(gchannels, r, eps, gaussian=gaussian)

    def forward(self, refs, logits, confidence=None, labels=None):
        """refs [B, C, H, W], logits [B, L, H, W]."""
        E0 = -logits if confidence is None else -logits * confidence
        extra = () if labels is None else (labels,)
        if (isinstance(self.W, BatchedAdjacency) and self.niters > 0 and E0.is_cuda and E0.dtype == torch.float32
                and not (torch.is_grad_enabled() and (E0.requires_grad or refs.requires_grad
                                                      or any(p.requires_grad for p in self.Mu.parameters())))):
            M = _compat_matrix(self.Mu, E0.shape[1], labels, E0.device)
            if M is not None:
                return _mean_field_nchw_fused(E0.contiguous(), refs, M, self.niters)      # already -E
        return -_mean_field_nchw(E0, lambda Q: self.W(self.Mu(Q, *extra), refs), self.niters)


class ijGuide(nn.Module):
    """Guide features (i, j) / sqrt(h^2 + w^2) / s_ij  (crf_module.py:116-123)."""

    def __init__(self, s_ij=.1, trainable=True):
        super().__init__()
        self.s_ij = _maybe_learnable(s_ij, trainable)

    def positions(self, x):
        bs, _, h, w = x.shape
        grid = np.mgrid[:h, :w] / np.sqrt(h ** 2 + w ** 2)
        return torch.from_numpy(grid).float().to(x.device)[None].expand(bs, -1, -1, -1) / self.s_ij

    def forward(self, x):
        return self.positions(x)


class ijrgbGuide(ijGuide):
    """Guide features (i, j)/s_ij ++ rgb/s_rgb (crf_module.py:106-114)."""

    def __init__(self, s_ij=.1, s_rgb=.1, trainable=True):
        super().__init__(s_ij, trainable)
        self.s_rgb = _maybe_learnable(s_rgb, trainable)

    def forward(self, x):
        return torch.cat([self.positions(x), x / self.s_rgb], dim=1)

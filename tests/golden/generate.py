#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF, run in the build container.

Container only: needs /root/reference.  The GPU box and the test-suite only ever read the
committed .npz files (plain arrays: inputs and expected outputs, no reference source text).

Two sources of truth are used, both the reference's own code:

1. The C++ lattice engine  crf/lattice/lite/permutohedral.h  compiled torch-free by
   oracle/build_ref.sh into oracle/_ref/libphl_ref.so  (splat / blur / slice, rows a8-a14 of
   SURVEY.md section 8a).
2. The Python callers  crf/crf_module.py, crf/gaussian_matrix.py  imported from
   /root/reference (rows a1-a6).  Two things that module needs at import time do not exist in
   this image and are NOT on the hot path:
     * ``guided_filter_pytorch`` (pip package used only by the guided-filter classes): an empty
       placeholder module is registered so that the import statement succeeds; none of its
       symbols is ever called here;
     * ``torch.utils.cpp_extension.load(... lattice.cpp)``: the shipped pybind wrapper no longer
       compiles on torch 2.10, so ``load`` is pointed at a ctypes wrapper
       around the same engine built in (1) -- i.e. ``lattice.filter`` is still the reference's
       own splat/blur/slice.

3. The unary cost volume  crf/depth.py:36-53 ``disparity_badness``  (SURVEY 8f-3), imported from
   /root/reference by file path.  Its module header imports ``cv2`` / ``cv2.ximgproc.guidedFilter``
   (absent from this image, never called by ``disparity_badness``): an empty placeholder module is
   registered so that the import statement succeeds.  oracle/costvol_oracle.py is required to equal
   the reference's output bit for bit (float64) on every stored case.

While generating, the script also PINS oracle/phl_oracle.c: every lattice case is required to
match the reference engine bit-for-bit (keys, replay offsets, weights, post-splat and
post-blur vertex values, output), including cases that grow the reference's hash table, where
the oracle's ``faithful_table`` mode reproduces the reference's stale-slot defect
(phl_oracle.c, table_lookup).  The summary is written to PIN_REPORT.json.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("PHL_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import phl_oracle as po  # noqa: E402


def sorted_by_key(keys, *arrays):
    """Vertex numbering is an implementation detail: store per-vertex data sorted by key."""
    order = np.lexsort(keys.T[::-1])
    return (keys[order],) + tuple(a[order] for a in arrays)


def lattice_case(name, n, d, vd, scale, seed, report, store=True, data=None):
    rng = np.random.default_rng(seed)
    recipe = None
    if data is None:
        ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        src = rng.standard_normal((n, vd)).astype(np.float32)
    else:
        ref, src, recipe = data
    R = po.reference_filter(src, ref, stages=True)
    grew = R["M"] >= 16383
    O = po.Oracle(ref, faithful_table=True)
    out, sd, bd = O.filter(src, stages=True)
    vid, w = O.replay()
    pinned = (O.M == R["M"] and np.array_equal(O.keys(), R["keys"]) and np.array_equal(vid, R["replay_vid"])
              and np.array_equal(w, R["replay_w"]) and np.array_equal(sd, R["splat"])
              and np.array_equal(bd, R["blur"]) and np.array_equal(out, R["out"]))
    n, d, vd = ref.shape[0], ref.shape[1], src.shape[1]
    entry = dict(case=name, n=n, d=d, vd=vd, scale=scale, seed=seed, M=int(R["M"]), table_grew=bool(grew),
                 oracle_faithful_bit_exact=bool(pinned))
    # clean-table oracle (what the HIP path is held to) versus the reference
    Oc = po.Oracle(ref)
    outc = Oc.filter(src)
    diff_rows = int((outc != R["out"]).any(1).sum())
    denom = np.maximum(np.abs(R["out"]), 1e-3 * np.abs(R["out"]).max())
    entry.update(clean_M=int(Oc.M), clean_rows_differing=diff_rows,
                 clean_rows_beyond_1e4=int(((np.abs(outc - R["out"]) / denom) > 1e-4).any(1).sum()),
                 clean_max_rel=float((np.abs(outc - R["out"]) / denom).max()))
    report.append(entry)
    assert pinned, f"oracle is not bit-exact against the reference engine on {name}"
    if not grew:
        assert diff_rows == 0 and Oc.M == R["M"]
    if store == "growth":
        # Above the reference's first table doubling (M >= 16383) its stale-slot defect (oracle/phl_oracle.c,
        # table_lookup) makes a few output rows differ from the defect-free algorithm.  Stored: the REFERENCE's
        # output, and the mask of rows where the oracle's faithful and clean modes differ (the rows the defect
        # reaches).  Per-vertex dumps are not stored (duplicate keys make "sorted by key" ambiguous).
        mask = (outc != out).any(1)
        assert np.array_equal(out, R["out"])
        entry["mask_rows"] = int(mask.sum())
        entry["mask_fraction"] = float(mask.mean())
        stored = dict(src=src, out=R["out"], M=np.int64(R["M"]), clean_M=np.int64(Oc.M), defect_mask=mask)
        if recipe is None:
            stored["ref"] = ref
        else:       # features are a deterministic function of an 8-bit image: store that (tests/_golden_util.py rebuilds ref)
            stored.update(recipe)
            from _golden_util import features_from_recipe
            assert np.array_equal(features_from_recipe(recipe), ref)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **stored)
    elif store:
        keys_s, splat_s, blur_s = sorted_by_key(R["keys"], R["splat"], R["blur"])
        np.savez_compressed(os.path.join(HERE, f"lattice_{name}.npz"), ref=ref, src=src, out=R["out"],
                            M=np.int64(R["M"]), keys_sorted=keys_s, splat_sorted=splat_s, blur_sorted=blur_s,
                            replay_w=R["replay_w"], replay_key=R["keys"][R["replay_vid"]])
    print(entry)


def tsukuba_case(vd, seed, sigma_c=0.08, sigma_p=0.03):
    """BASELINE configs[0] geometry: the reference's Tsukuba left image (Experiments/imL.png, 384x288), 5-D
    bilateral features (rgb / sigma_c, ij / diag / sigma_p) as Experiments/DenseCrf.ipynb cell 9 builds them,
    at the (.08, .03) setting of BASELINE.md (M/n ~ 0.21, M ~ 23 k: above the first table doubling).
    Values: 8-bit-quantised uniform noise (compresses; any values pin the arithmetic)."""
    from PIL import Image
    from _golden_util import features_from_recipe

    img = np.asarray(Image.open(os.path.join(REFERENCE, "Experiments", "imL.png")).convert("RGB"))
    recipe = dict(img_u8=img, sigma_c=np.float64(sigma_c), sigma_p=np.float64(sigma_p))
    ref = features_from_recipe(recipe)
    rng = np.random.default_rng(seed)
    src = (rng.integers(0, 256, size=(ref.shape[0], vd)) / 255.0).astype(np.float32)
    return ref, src, recipe


# ------------------------------------------------------------------------------------------------
def import_reference_python():
    """Import /root/reference/crf with lattice.filter bound to the reference engine (see header)."""
    import torch
    import torch.utils.cpp_extension as cpp_ext

    placeholder = types.ModuleType("guided_filter_pytorch")
    sub = types.ModuleType("guided_filter_pytorch.guided_filter")
    # constructible (CRFasRNN.__init__ builds its default guided-filter W before the lattice W is substituted),
    # never called
    init = lambda self, *a, **k: torch.nn.Module.__init__(self)  # noqa: E731
    sub.GuidedFilter = type("GuidedFilter", (torch.nn.Module,), {"__init__": init})
    sub.BoxFilter = type("BoxFilter", (torch.nn.Module,), {"__init__": init})
    placeholder.guided_filter = sub
    sys.modules["guided_filter_pytorch"] = placeholder
    sys.modules["guided_filter_pytorch.guided_filter"] = sub

    import _ref_lattice_shim

    cpp_ext.load = lambda *a, **k: _ref_lattice_shim
    sys.path.insert(0, REFERENCE)
    import crf.crf_module as crf_module
    import crf.gaussian_matrix as gm

    assert gm.latticefilter is _ref_lattice_shim.filter
    return crf_module, gm


def read_image(path):
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB")).astype(np.float64) / 255.0


def disparity_badness(img1, img2, max_disp, ws=9):
    """Caller-side input generator (shifted absolute difference + ws x ws box aggregate), after
    crf/depth.py:36-53 with max_disp as a parameter.  It only produces the INPUT E_0."""
    from scipy import ndimage

    h, w, _ = img1.shape
    padded = np.pad(img2, ((0, 0), (max_disp, 0), (0, 0)), mode="constant")
    out = np.zeros((h, w, max_disp))
    for i in range(max_disp):
        out[:, :, i] = np.abs(img1 - padded[:, max_disp - i:w + max_disp - i]).sum(2)
    return ndimage.convolve(out, np.ones((ws, ws, 1)))


def python_layer_cases(report):
    import torch

    torch.manual_seed(0)
    crf_module, gm = import_reference_python()

    # ---- a1/a2/a3: Tsukuba crop, 5-D bilateral features, Charbonnier compat, mean field -------
    imL = read_image(os.path.join(REFERENCE, "Experiments", "imL.png"))
    imR = read_image(os.path.join(REFERENCE, "Experiments", "imR.png"))
    L, sigma_c, sigma_p, gamma = 16, 0.1, 0.1, 3
    full = disparity_badness(imL, imR, L)
    r0, c0, h, w = 96, 150, 48, 64
    E0 = torch.from_numpy(full[r0:r0 + h, c0:c0 + w].reshape(-1, L)).float()
    H, W_ = imL.shape[:2]
    position = np.mgrid[:H, :W_].transpose((1, 2, 0)) / np.sqrt(H ** 2 + W_ ** 2)
    refimg = np.zeros((h, w, 5))
    refimg[..., :3] = imL[r0:r0 + h, c0:c0 + w] / sigma_c
    refimg[..., 3:] = position[r0:r0 + h, c0:c0 + w] / sigma_p
    flat_ref = torch.from_numpy(refimg.reshape(h * w, -1).astype(np.float32))
    labels = torch.arange(L).float()
    Mu = crf_module.compatibility_matrix(lambda a, b: crf_module.charbonneir(a, b, gamma), labels)
    Wop = gm.LatticeGaussian(flat_ref)
    with torch.no_grad():
        Q0 = torch.softmax(-E0, dim=1)
        WQ0 = Wop @ Q0
        Q1 = crf_module.mean_field_infer(E0, Wop, Mu, 1)
        Q5 = crf_module.mean_field_infer(E0, Wop, Mu, 5)
    np.savez_compressed(os.path.join(HERE, "meanfield_tsukuba_crop.npz"), E0=E0.numpy(), ref=flat_ref.numpy(),
                        labels=labels.numpy(), Mu=Mu.numpy(), gamma=np.float32(gamma), WQ0=WQ0.numpy(),
                        Q1=Q1.numpy(), Q5=Q5.numpy(), disp1=(Q1 @ labels).numpy(), disp5=(Q5 @ labels).numpy(),
                        h=np.int64(h), w=np.int64(w))
    report.append(dict(case="meanfield_tsukuba_crop", n=h * w, L=L, d=5))

    # ---- a4: LatticeFilter backward (grad_src, grad_ref) ------------------------------------
    for name, n, d, Lc, scale in [("grad_n80_d3_L2", 80, 3, 2, 1.0), ("grad_n2000_d5_L4", 2000, 5, 4, 4.0)]:
        g_ = torch.Generator().manual_seed(1234)
        ref = (torch.rand(n, d, generator=g_) * scale).requires_grad_(True)
        src = torch.randn(n, Lc, generator=g_).requires_grad_(True)
        gout = torch.randn(n, Lc, generator=g_)
        out = gm.LatticeFilter.apply(src, ref)
        out.backward(gout)
        # src-only branch (ref does not need grad): one filter of g
        src2 = src.detach().clone().requires_grad_(True)
        out2 = gm.LatticeFilter.apply(src2, ref.detach())
        try:
            out2.backward(gout)
            src_only = src2.grad.numpy()
            src_only_raises = ""
        except UnboundLocalError as e:
            # reference defect: gaussian_matrix.py:467 prints a timing array `s` that only the
            # ref-grad branch (:449) defines, so the src-only branch (:445-446) always raises after
            # computing grad_source = latticefilter(g, ref).  The intended value is recorded.
            src_only = gm.latticefilter(gout, ref.detach()).numpy()
            src_only_raises = repr(e)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), ref=ref.detach().numpy(), src=src.detach().numpy(),
                            gout=gout.numpy(), out=out.detach().numpy(), grad_src=src.grad.numpy(),
                            grad_ref=ref.grad.numpy(), grad_src_only=src_only)
        report.append(dict(case=name, n=n, d=d, L=Lc, reference_src_only_backward_raises=src_only_raises))

    # ---- a5: BatchedAdjacency on NCHW tensors (strided views, "- src") -----------------------
    g_ = torch.Generator().manual_seed(99)
    srcb = torch.randn(2, 4, 12, 16, generator=g_)
    guide = torch.rand(2, 5, 12, 16, generator=g_) * 3.0
    with torch.no_grad():
        outb = gm.BatchedAdjacency(num_threads=2)(srcb, guide)
    np.savez_compressed(os.path.join(HERE, "batched_adjacency.npz"), src=srcb.numpy(), guide=guide.numpy(),
                        out=outb.numpy())
    report.append(dict(case="batched_adjacency", shape=list(srcb.shape)))

    # ---- a6: Laplacian operators -----------------------------------------------------------
    g_ = torch.Generator().manual_seed(7)
    n, d, k = 1500, 3, 3
    ref = torch.rand(n, d, generator=g_) * 2.5
    U = torch.randn(n, k, generator=g_)
    store = dict(ref=ref.numpy(), U=U.numpy())
    with torch.no_grad():
        for norm in (True, False):
            op = gm.RbfLaplacian(ref, normalize=norm)
            store[f"rbf_D"] = op.D.numpy()
            store[f"rbf_{'norm' if norm else 'unnorm'}"] = (op @ U).numpy()
        for norm in ("sym", "right", "none"):
            op = gm.RbfLaplacianC(ref, normalize=norm)
            store["rbfc_D"] = op.D.numpy()
            store[f"rbfc_{norm}"] = (op @ U).numpy()
    np.savez_compressed(os.path.join(HERE, "laplacians.npz"), **store)
    report.append(dict(case="laplacians", n=n, d=d, k=k))


def mean_field_wide_cases(report):
    """Vectors for the fused compatibility + softmax kernel (L a multiple of 32) and for the NCHW CRFasRNN path,
    both from the reference's own Python (crf/crf_module.py) over the reference engine."""
    import torch

    torch.manual_seed(1)
    # the reference's BatchedAdjacency fork()s a process pool (gaussian_matrix.py:370-377): OpenMP worker threads
    # started by an earlier matmul would leave the forked children waiting forever
    torch.set_num_threads(1)
    crf_module, gm = import_reference_python()
    imL = read_image(os.path.join(REFERENCE, "Experiments", "imL.png"))
    imR = read_image(os.path.join(REFERENCE, "Experiments", "imR.png"))

    # ---- flat mean field, L = 32 (crf_module.py:41-53) ------------------------------------------
    L, sigma_c, sigma_p, gamma = 32, 0.1, 0.1, 3
    full = disparity_badness(imL, imR, L)
    r0, c0, h, w = 120, 180, 40, 56
    E0 = torch.from_numpy(full[r0:r0 + h, c0:c0 + w].reshape(-1, L)).float()
    H, W_ = imL.shape[:2]
    position = np.mgrid[:H, :W_].transpose((1, 2, 0)) / np.sqrt(H ** 2 + W_ ** 2)
    refimg = np.zeros((h, w, 5))
    refimg[..., :3] = imL[r0:r0 + h, c0:c0 + w] / sigma_c
    refimg[..., 3:] = position[r0:r0 + h, c0:c0 + w] / sigma_p
    flat_ref = torch.from_numpy(refimg.reshape(h * w, -1).astype(np.float32))
    labels = torch.arange(L).float()
    Mu = crf_module.compatibility_matrix(lambda a, b: crf_module.charbonneir(a, b, gamma), labels)
    Wop = gm.LatticeGaussian(flat_ref)
    with torch.no_grad():
        Q1 = crf_module.mean_field_infer(E0, Wop, Mu, 1)
        Q5 = crf_module.mean_field_infer(E0, Wop, Mu, 5)
    np.savez_compressed(os.path.join(HERE, "meanfield_tsukuba_L32.npz"), E0=E0.numpy(), ref=flat_ref.numpy(),
                        labels=labels.numpy(), Mu=Mu.numpy(), gamma=np.float32(gamma), Q1=Q1.numpy(), Q5=Q5.numpy(),
                        disp1=(Q1 @ labels).numpy(), disp5=(Q5 @ labels).numpy(), h=np.int64(h), w=np.int64(w))
    report.append(dict(case="meanfield_tsukuba_L32", n=h * w, L=L, d=5))

    # ---- CRFasRNN, NCHW, lattice W (crf_module.py:66-104 with BatchedAdjacency as self.W) -------
    g_ = torch.Generator().manual_seed(321)
    bs, L2, hh, ww = 2, 32, 20, 24
    img = torch.from_numpy(np.stack([imL[60:60 + hh, 100:100 + ww], imL[150:150 + hh, 220:220 + ww]]).transpose(0, 3, 1, 2)).float()
    logits = torch.randn(bs, L2, hh, ww, generator=g_) * 2.0
    lab = torch.arange(L2).float()
    net = crf_module.CRFasRNN(crf_module.charb(3.0), niters=3)
    net.W = gm.BatchedAdjacency(num_threads=2)          # the lattice alternative the reference imports (:5) but does not wire in
    guide = crf_module.ijrgbGuide(trainable=False)
    with torch.no_grad():
        refs = guide(img)
        out = net(refs, logits, labels=lab)
        conf = torch.rand(bs, 1, hh, ww, generator=g_) + 0.5
        out_conf = net(refs, logits, confidence=conf, labels=lab)
    np.savez_compressed(os.path.join(HERE, "crfasrnn_nchw.npz"), img=img.numpy(), logits=logits.numpy(), refs=refs.numpy(),
                        labels=lab.numpy(), gamma=np.float32(3.0), niters=np.int64(3), out=out.numpy(),
                        confidence=conf.numpy(), out_conf=out_conf.numpy())
    report.append(dict(case="crfasrnn_nchw", shape=list(logits.shape), niters=3))


def round4_cases(report):
    """Round 4: (i) a reference-autograd vector INSIDE the fused backward's envelope at a width that takes k_splat_wide
    (d = 5, L = 64, image-like features: 48 x 64 pixels), (ii) the flat mean field at the notebook's own label count
    L = 384 // 6 = 64 (DenseCrf.ipynb:95-115, crf/depth.py:40) on a Tsukuba crop, 1 and 5 iterations.  Inputs of (i) are
    stored as float16 (the values are rounded to float16 first, so the stored inputs are exact)."""
    import torch

    torch.manual_seed(2)
    torch.set_num_threads(1)
    crf_module, gm = import_reference_python()
    imL = read_image(os.path.join(REFERENCE, "Experiments", "imL.png"))
    imR = read_image(os.path.join(REFERENCE, "Experiments", "imR.png"))
    H, W_ = imL.shape[:2]
    position = np.mgrid[:H, :W_].transpose((1, 2, 0)) / np.sqrt(H ** 2 + W_ ** 2)

    # (i) gradient through the lattice, image-like features
    r0, c0, h, w, Lc = 130, 200, 48, 64, 64
    refimg = np.zeros((h, w, 5))
    refimg[..., :3] = imL[r0:r0 + h, c0:c0 + w] / 0.125
    refimg[..., 3:] = position[r0:r0 + h, c0:c0 + w] / 0.01
    rng = np.random.default_rng(404)
    ref_np = refimg.reshape(h * w, 5).astype(np.float32)
    src_np = rng.random((h * w, Lc)).astype(np.float16).astype(np.float32)
    gout_np = rng.standard_normal((h * w, Lc)).astype(np.float16).astype(np.float32)
    ref = torch.from_numpy(ref_np).requires_grad_(True)
    src = torch.from_numpy(src_np).requires_grad_(True)
    out = gm.LatticeFilter.apply(src, ref)
    out.backward(torch.from_numpy(gout_np))
    np.savez_compressed(os.path.join(HERE, "grad_image_48x64_d5_L64.npz"), ref=ref_np, src_f16=src_np.astype(np.float16),
                        gout_f16=gout_np.astype(np.float16), grad_src=src.grad.numpy(), grad_ref=ref.grad.numpy(),
                        h=np.int64(h), w=np.int64(w))
    report.append(dict(case="grad_image_48x64_d5_L64", n=h * w, d=5, L=Lc))

    # (ii) flat mean field at L = 64
    L, sigma_c, sigma_p, gamma = 64, 0.1, 0.1, 3
    full = disparity_badness(imL, imR, L)
    r0, c0, h, w = 150, 240, 32, 48
    E0 = torch.from_numpy(full[r0:r0 + h, c0:c0 + w].reshape(-1, L)).float()
    refimg = np.zeros((h, w, 5))
    refimg[..., :3] = imL[r0:r0 + h, c0:c0 + w] / sigma_c
    refimg[..., 3:] = position[r0:r0 + h, c0:c0 + w] / sigma_p
    flat_ref = torch.from_numpy(refimg.reshape(h * w, -1).astype(np.float32))
    labels = torch.arange(L).float()
    Mu = crf_module.compatibility_matrix(lambda a, b: crf_module.charbonneir(a, b, gamma), labels)
    Wop = gm.LatticeGaussian(flat_ref)
    with torch.no_grad():
        Q1 = crf_module.mean_field_infer(E0, Wop, Mu, 1)
        Q5 = crf_module.mean_field_infer(E0, Wop, Mu, 5)
    np.savez_compressed(os.path.join(HERE, "meanfield_tsukuba_L64.npz"), E0=E0.numpy(), ref=flat_ref.numpy(),
                        labels=labels.numpy(), gamma=np.float32(gamma), Q1=Q1.numpy(), Q5=Q5.numpy(),
                        disp1=(Q1 @ labels).numpy(), disp5=(Q5 @ labels).numpy(), h=np.int64(h), w=np.int64(w))
    report.append(dict(case="meanfield_tsukuba_L64", n=h * w, L=L, d=5))


def import_reference_depth():
    import importlib.util

    cv2 = types.ModuleType("cv2")
    cv2.ximgproc = types.ModuleType("cv2.ximgproc")
    cv2.ximgproc.guidedFilter = None          # placeholder; disparity_badness never touches it
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("cv2.ximgproc", cv2.ximgproc)
    spec = importlib.util.spec_from_file_location("reference_crf_depth", os.path.join(REFERENCE, "crf", "depth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cost_volume_cases(report):
    """Reference outputs of disparity_badness (max_disp = w // 6 as it fixes it) + pin of the numpy oracle."""
    from oracle import costvol_oracle as co

    ref_depth = import_reference_depth()
    imL = read_image(os.path.join(REFERENCE, "Experiments", "imL.png"))
    imR = read_image(os.path.join(REFERENCE, "Experiments", "imR.png"))
    rng = np.random.default_rng(77)
    cases = [
        ("tsukuba_crop_ad9", imL[100:140, 120:216], imR[100:140, 120:216], 9, "AD"),          # 40 x 96 x 3 -> L = 16
        ("tsukuba_edge_sd5", imL[:21, :66], imR[:21, :66], 5, "SD"),                          # touches two borders, L = 11
        ("random_nprod3_c1", rng.random((13, 31, 1)), rng.random((13, 31, 1)), 3, "nprod"),   # ragged tile, 1 channel
        ("random_ad17_c4", rng.random((19, 50, 4)), rng.random((19, 50, 4)), 17, "AD"),       # widest window, 4 channels
        ("tiny_ad9", rng.random((3, 12, 3)), rng.random((3, 12, 3)), 9, "AD"),               # window larger than the image
    ]
    for name, a, b, ws, crit in cases:
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        want = ref_depth.disparity_badness(a, b, ws, getattr(ref_depth, crit))
        got = co.disparity_badness(a, b, ws, crit)
        pinned = bool(want.shape == got.shape and np.array_equal(want, got))
        assert pinned, f"cost-volume oracle differs from the reference on {name}"
        np.savez_compressed(os.path.join(HERE, f"costvol_{name}.npz"), img1=a, img2=b, window=np.int64(ws),
                            criterion=np.array(crit), out=want)
        report.append(dict(case=f"costvol_{name}", shape=list(want.shape), window=ws, criterion=crit, oracle_bit_exact=pinned))


def growth_cases(report):
    # cases that grow the reference's hash table (M >= 16383, as every BASELINE GPU config does): the
    # reference's output is stored together with the mask of rows its stale-slot defect reaches
    lattice_case("growth_n20000_d5_vd4", 20000, 5, 4, 8.0, 21, report=report, store="growth")
    lattice_case("growth_tsukuba_384x288_vd4", 0, 5, 4, 0.0, 24, report=report, store="growth", data=tsukuba_case(4, 24))
    # pin-only (too big to store)
    for args in [("grow_n60000_d5_vd3", 60000, 5, 3, 6.0, 22), ("grow_n200000_d3_vd2", 200000, 3, 2, 40.0, 23)]:
        lattice_case(*args, report=report, store=False)


def round4b_cases(report):
    """Round 4, second half: the flat mean field at L = 256 (the label count of BASELINE's C2 / C3 configurations, where the
    compatibility product runs on the bf16 matrix cores with split operands) on a 16 x 24 Tsukuba crop = three 128-pixel
    tiles, 1 and 5 iterations of the reference (crf_module.py:41-53 over the reference engine).  E_0 is rounded to float16
    first and stored as float16, so the stored input is exact."""
    import torch

    torch.manual_seed(3)
    torch.set_num_threads(1)
    crf_module, gm = import_reference_python()
    imL = read_image(os.path.join(REFERENCE, "Experiments", "imL.png"))
    imR = read_image(os.path.join(REFERENCE, "Experiments", "imR.png"))
    H, W_ = imL.shape[:2]
    position = np.mgrid[:H, :W_].transpose((1, 2, 0)) / np.sqrt(H ** 2 + W_ ** 2)
    L, sigma_c, sigma_p, gamma = 256, 0.1, 0.1, 3
    r0, c0, h, w = 150, 300, 16, 24
    full = disparity_badness(imL, imR, L)
    E0_np = full[r0:r0 + h, c0:c0 + w].reshape(-1, L).astype(np.float16).astype(np.float32)
    E0 = torch.from_numpy(E0_np)
    refimg = np.zeros((h, w, 5))
    refimg[..., :3] = imL[r0:r0 + h, c0:c0 + w] / sigma_c
    refimg[..., 3:] = position[r0:r0 + h, c0:c0 + w] / sigma_p
    flat_ref = torch.from_numpy(refimg.reshape(h * w, -1).astype(np.float32))
    labels = torch.arange(L).float()
    Mu = crf_module.compatibility_matrix(lambda a, b: crf_module.charbonneir(a, b, gamma), labels)
    Wop = gm.LatticeGaussian(flat_ref)
    with torch.no_grad():
        Q1 = crf_module.mean_field_infer(E0, Wop, Mu, 1)
        Q5 = crf_module.mean_field_infer(E0, Wop, Mu, 5)
    np.savez_compressed(os.path.join(HERE, "meanfield_tsukuba_L256.npz"), E0_f16=E0_np.astype(np.float16), ref=flat_ref.numpy(),
                        labels=labels.numpy(), gamma=np.float32(gamma), Q1=Q1.numpy(), Q5=Q5.numpy(),
                        disp1=(Q1 @ labels).numpy(), disp5=(Q5 @ labels).numpy(), h=np.int64(h), w=np.int64(w))
    report.append(dict(case="meanfield_tsukuba_L256", n=h * w, L=L, d=5))


def main():
    if sys.argv[1:] == ["round4b"]:
        assert po.build_reference(), "reference engine not built"
        report = json.load(open(os.path.join(HERE, "PIN_REPORT.json")))
        report = [r for r in report if r.get("case") not in ("meanfield_tsukuba_L256",)]
        round4b_cases(report)
        with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
            json.dump(report, f, indent=1)
        return
    if sys.argv[1:] == ["costvol"]:          # add the cost-volume vectors without regenerating the rest
        report = json.load(open(os.path.join(HERE, "PIN_REPORT.json")))
        report = [r for r in report if not str(r.get("case", "")).startswith("costvol_")]
        cost_volume_cases(report)
        with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
            json.dump(report, f, indent=1)
        print("wrote", sorted(x for x in os.listdir(HERE) if x.startswith("costvol_")))
        return
    if sys.argv[1:] == ["meanfield2"]:
        assert po.build_reference(), "reference engine not built"
        report = json.load(open(os.path.join(HERE, "PIN_REPORT.json")))
        report = [r for r in report if r.get("case") not in ("meanfield_tsukuba_L32", "crfasrnn_nchw")]
        mean_field_wide_cases(report)
        with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
            json.dump(report, f, indent=1)
        return
    if sys.argv[1:] == ["round4"]:
        assert po.build_reference(), "reference engine not built"
        report = json.load(open(os.path.join(HERE, "PIN_REPORT.json")))
        report = [r for r in report if r.get("case") not in ("grad_image_48x64_d5_L64", "meanfield_tsukuba_L64")]
        round4_cases(report)
        with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
            json.dump(report, f, indent=1)
        return
    if sys.argv[1:] == ["growth"]:           # add / refresh the stored table-growth cases only
        po.build_oracle(force=True)
        assert po.build_reference(), "reference engine not built"
        report = json.load(open(os.path.join(HERE, "PIN_REPORT.json")))
        report = [r for r in report if not str(r.get("case", "")).startswith("grow")]
        growth_cases(report)
        with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
            json.dump(report, f, indent=1)
        return
    po.build_oracle(force=True)
    assert po.build_reference(), "reference engine not built"
    report = []
    # (name, n, d, vd, feature scale, seed) -- stored
    for args in [("n64_d2_vd1", 64, 2, 1, 3.0, 11), ("n500_d1_vd2", 500, 1, 2, 10.0, 12),
                 ("n2000_d3_vd3", 2000, 3, 3, 5.0, 13), ("n2000_d5_vd16", 2000, 5, 16, 4.0, 14),
                 ("n3000_d8_vd5", 3000, 8, 5, 2.0, 15), ("n4096_d5_vd64", 4096, 5, 64, 2.0, 16),
                 ("n1000_d2_vd7_wide", 1000, 2, 7, 300.0, 17)]:
        lattice_case(*args, report=report)
    growth_cases(report)
    python_layer_cases(report)
    mean_field_wide_cases(report)
    round4_cases(report)
    round4b_cases(report)
    cost_volume_cases(report)
    with open(os.path.join(HERE, "PIN_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", sorted(x for x in os.listdir(HERE) if x.endswith(".npz")))


if __name__ == "__main__":
    main()

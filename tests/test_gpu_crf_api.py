"""GPU: the mirrored crf.* Python surface (rows a1-a6 of SURVEY.md 8a) against vectors produced
by the reference's own Python + C++ in the build container (tests/golden/generate.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # north_star: disparity maps within 1e-4 relative per pixel


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * np.abs(b).max())).max())


def test_mean_field_tsukuba_crop(golden_dir):
    from crf.crf_module import charbonneir, compatibility_matrix, mean_field_infer
    from crf.gaussian_matrix import LatticeGaussian

    g = np.load(os.path.join(golden_dir, "meanfield_tsukuba_crop.npz"))
    dev = torch.device("cuda")
    E0 = torch.from_numpy(g["E0"]).to(dev)
    ref = torch.from_numpy(g["ref"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    Mu = compatibility_matrix(lambda a, b: charbonneir(a, b, float(g["gamma"])), labels)
    assert rel(Mu.cpu().numpy(), g["Mu"]) <= 1e-6
    W = LatticeGaussian(ref)
    WQ0 = W @ torch.softmax(-E0, dim=1)
    assert rel(WQ0.cpu().numpy(), g["WQ0"]) <= RTOL
    for it, key in ((1, "1"), (5, "5")):
        Q = mean_field_infer(E0, W, Mu, it)
        eq = rel(Q.cpu().numpy(), g["Q" + key])
        disp = (Q @ labels).cpu().numpy()
        # north star: "output disparity maps match the reference CPU path within 1e-4 relative per pixel"
        ed = float((np.abs(disp - g["disp" + key]) / np.maximum(np.abs(g["disp" + key]), 1e-2)).max())
        print(f"[measured] mean field, {it} iteration(s): Q rel {eq:.2e}, disparity rel per pixel {ed:.2e}")
        # Q = softmax(-E): a relative error in Q is an ABSOLUTE error in E (values 10..50 here), so 1e-4 on a
        # small probability is 1e-5..1e-6 relative on the energy the filter produced; measured 1.2e-4
        assert eq <= 5e-4
        assert ed <= 2.5e-5          # north star: 1e-4 per pixel; bound = 10x the measured 2.4e-6
    # CPU tensors in, CPU tensors out (the notebook runs on device('cpu'))
    Qc = mean_field_infer(E0.cpu(), LatticeGaussian(ref.cpu()), Mu.cpu(), 1)
    assert not Qc.is_cuda and rel(Qc.numpy(), g["Q1"]) <= 5e-4


@pytest.mark.parametrize("name", ["grad_n80_d3_L2", "grad_n2000_d5_L4"])
def test_lattice_filter_backward(golden_dir, name):
    from crf.gaussian_matrix import LatticeFilter

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dev = torch.device("cuda")
    ref = torch.from_numpy(g["ref"]).to(dev).requires_grad_(True)
    src = torch.from_numpy(g["src"]).to(dev).requires_grad_(True)
    gout = torch.from_numpy(g["gout"]).to(dev)
    out = LatticeFilter.apply(src, ref)
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() <= 1e-5 * np.abs(g["out"]).max()
    out.backward(gout)
    assert rel(src.grad.cpu().numpy(), g["grad_src"]) <= RTOL
    eg = rel(ref.grad.cpu().numpy(), g["grad_ref"])
    print(f"[measured] {name}: grad_ref rel {eg:.2e}")
    assert eg <= 5e-4      # reference's own gradcheck rtol (gaussian_matrix.py:516); a sum of 4L cancelling products
    src2 = src.detach().clone().requires_grad_(True)
    LatticeFilter.apply(src2, ref.detach()).backward(gout)
    assert rel(src2.grad.cpu().numpy(), g["grad_src_only"]) <= RTOL


def test_batched_adjacency_nchw(golden_dir):
    from crf.gaussian_matrix import BatchedAdjacency

    g = np.load(os.path.join(golden_dir, "batched_adjacency.npz"))
    src = torch.from_numpy(g["src"]).cuda()
    guide = torch.from_numpy(g["guide"]).cuda()
    out = BatchedAdjacency(num_threads=2)(src, guide)
    assert out.shape == src.shape and rel(out.cpu().numpy(), g["out"]) <= RTOL


def test_laplacians(golden_dir):
    from crf.gaussian_matrix import RbfLaplacian, RbfLaplacianC

    g = np.load(os.path.join(golden_dir, "laplacians.npz"))
    ref = torch.from_numpy(g["ref"]).cuda()
    U = torch.from_numpy(g["U"]).cuda()
    op = RbfLaplacian(ref, normalize=True)
    assert rel(op.D.cpu().numpy(), g["rbf_D"]) <= RTOL
    assert rel((op @ U).cpu().numpy(), g["rbf_norm"]) <= RTOL
    assert rel((RbfLaplacian(ref, normalize=False) @ U).cpu().numpy(), g["rbf_unnorm"]) <= RTOL
    for mode in ("sym", "right", "none"):
        opc = RbfLaplacianC(ref, normalize=mode)
        assert rel(opc.D.cpu().numpy(), g["rbfc_D"]) <= RTOL
        el = rel((opc @ U).cpu().numpy(), g["rbfc_" + mode])
        print(f"[measured] RbfLaplacianC {mode}: {el:.2e}")
        assert el <= RTOL * 3   # "- U" cancellation


def test_crf_as_rnn_lattice_runs():
    from crf.crf_module import CRFasRNN, charb, ijrgbGuide

    torch.manual_seed(0)
    net = CRFasRNN(charb(3.0), niters=2, lattice=True).cuda()
    img = torch.rand(2, 3, 24, 32, device="cuda")
    logits = torch.randn(2, 8, 24, 32, device="cuda", requires_grad=True)
    refs = ijrgbGuide(trainable=False)(img)
    out = net(refs, logits)
    assert out.shape == logits.shape and torch.isfinite(out).all()
    out.sum().backward()
    assert torch.isfinite(logits.grad).all()


@pytest.mark.parametrize("n,L", [(1000, 16), (4097, 256), (300, 300), (257, 1000), (64, 3), (50, 2048)])
def test_fused_softmax_neg_add_and_expected_value(n, L):
    """phl.softmax_neg_add == softmax(-(E0 + G)) of crf_module.py:49-52; fp32 tolerance (different
    exp / reduction order than torch), rows sum to 1."""
    import phl

    g = torch.Generator(device="cuda").manual_seed(n + L)
    E0 = torch.rand((n, L), device="cuda", generator=g) * 20
    G = torch.randn((n, L), device="cuda", generator=g) * 5
    want = torch.softmax(-(E0.double() + G.double()), dim=1)
    got = phl.softmax_neg_add(E0, G)
    # the fp32 sum E0+G carries ~1e-6 absolute rounding into the exponent, same as torch's fp32 path
    assert float((got.double() - want).abs().max()) <= 2e-6
    ref32 = torch.softmax(-(E0 + G), dim=1)
    big = ref32 > 1e-20
    assert float(((got - ref32).abs() / ref32.clamp_min(1e-30))[big].max()) <= 1e-5
    assert float((got.sum(1) - 1).abs().max()) <= 1e-5
    assert float((phl.softmax_neg_add(E0).double() - torch.softmax(-E0.double(), 1)).abs().max()) <= 2e-6
    # row-padded views take the same path
    pad = torch.zeros((n, L + 4), device="cuda")
    pad[:, :L] = E0
    assert torch.equal(phl.softmax_neg_add(pad[:, :L], G), got)
    labels = torch.arange(L, dtype=torch.float32, device="cuda") * 0.5
    ev = phl.expected_value(got, labels)
    assert float((ev.double() - got.double() @ labels.double()).abs().max()) <= 1e-4 * float(labels.max() + 1)


def test_torch_cpp_extension_binding():
    """lib/lattice_ext.so: pybind `filter(src, ref)` with the reference's signature
    (crf/lattice/lite/lattice.cpp:6-15) forwarding to phl_filter_once."""
    import importlib.util
    import os

    import phl

    path = os.path.join(os.path.dirname(phl.LIB_PATH), "lattice_ext.so")
    if not os.path.exists(path):
        pytest.skip("lattice_ext.so not built")
    spec = importlib.util.spec_from_file_location("lattice_ext", path)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    src = torch.rand(3000, 12, device="cuda")
    ref = torch.rand(3000, 5, device="cuda") * 4
    a = ext.filter(src, ref)
    b = phl.filter(src, ref)
    assert a.is_cuda and torch.equal(a, b)
    c = ext.filter(src.cpu(), ref.cpu())                    # CPU tensors in, CPU tensor out
    assert not c.is_cuda and torch.equal(c, b.cpu())
    with pytest.raises(RuntimeError, match="Incompatible shapes"):
        ext.filter(src[:10], ref)


def test_notebook_flow_end_to_end_on_a_synthetic_pair():
    """examples/stereo_crf.py: cost volume on the device -> lattice -> 5 mean-field iterations -> expected
    disparity.  The CRF estimate must beat the window-sweep estimate it starts from."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "stereo_crf.py")
    spec = importlib.util.spec_from_file_location("stereo_crf_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    err_wta, err_crf = mod.run(h=120, w=240, iters=5, quiet=True)
    assert err_crf < err_wta and err_crf < 0.5, (err_wta, err_crf)

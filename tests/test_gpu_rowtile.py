"""GPU: row-band decomposition with the HIP engine, all ranks played on one GPU (loopback), against
the single-lattice HIP filter; checks ghost-vertex insertion (phl_add_vertices) on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * np.abs(b).max())).max())


@pytest.mark.parametrize("world,H,W,L", [(2, 128, 96, 16), (4, 256, 64, 32), (8, 512, 48, 8)])
def test_bands_match_single_lattice_on_gpu(world, H, W, L):
    import phl
    from phl import rowtile
    from test_rowtile_cpu import make_image

    feat, src = make_image(H, W, L, sigma_xy=3.0)
    dev = torch.device("cuda")
    s = torch.from_numpy(src).to(dev)
    want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev)).filter(s).cpu().numpy()
    got, bands = rowtile.simulate(feat, s, world, phl.Lattice, dev)
    assert rel(got.cpu().numpy(), want) <= RTOL
    for b in bands:
        assert b.M >= b.eng.M_local and (b.M > b.eng.M_local or world == 1)


def test_add_vertices_semantics():
    import phl
    from oracle import phl_oracle as po

    rng = np.random.default_rng(2)
    ref = (rng.random((3000, 3), dtype=np.float32) * 4).astype(np.float32)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    O = po.Oracle(ref)
    keys = L.keys()
    M0 = L.M
    q = np.concatenate([keys[10:20], keys[:3] + np.int16(100), keys[50:52]])       # existing, new, existing
    ids = L.add_vertices(q)
    oids = O.add_vertices(q)
    assert np.array_equal(ids, oids) and L.M == O.M == M0 + 3 and L.M_local == M0
    assert np.array_equal(ids[:10], np.arange(10, 20)) and np.array_equal(ids[10:13], np.arange(M0, M0 + 3))
    assert np.array_equal(L.keys(), O.keys()) and np.array_equal(L.neighbors(), O.neighbors())
    src = rng.standard_normal((3000, 8)).astype(np.float32)
    a = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
    assert np.array_equal(a.view(np.uint32), O.filter(src).view(np.uint32))
    v = L.splat(torch.from_numpy(src).cuda())
    assert v.shape[0] == M0 + 3 and float(v[M0:].abs().max()) == 0.0               # ghosts receive nothing locally

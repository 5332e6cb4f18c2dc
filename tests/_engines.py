"""Test-only lattice engine with the phl.Lattice stage surface, backed by the CPU oracle, so that
the row-band decomposition and its exchange can run under gloo without a GPU."""
import numpy as np
import torch

from oracle import phl_oracle as po


class OracleEngine:
    def __init__(self, ref):
        self._o = po.Oracle(np.ascontiguousarray(ref.cpu().numpy(), np.float32))
        self.n, self.d = self._o.n, self._o.d

    @property
    def M(self):
        return self._o.M

    def keys(self):
        return self._o.keys()

    def replay(self):
        return self._o.replay()

    def add_vertices(self, keys):
        return self._o.add_vertices(keys)

    def splat(self, src):
        return torch.from_numpy(self._o.splat(src.cpu().numpy()))

    def blur(self, vert):
        return torch.from_numpy(self._o.blur(vert.cpu().numpy()))

    def slice(self, vert, out=None):
        res = torch.from_numpy(self._o.slice(vert.cpu().numpy()))
        if out is not None:
            out.copy_(res)
            return out
        return res


class _NumpyBand:
    """A band's lattice as plain arrays (restriction of the whole image's faithful CPU lattice): splat / blur / slice in
    numpy float32, following the reference's expressions (permutohedral.h:454-455, :526, :480)."""

    def __init__(self, keys, vid, w, nbr, n_own):
        self._keys, self._vid, self._w, self._nbr = keys, vid, w, nbr
        self.n, self.d = vid.shape[0], keys.shape[1]
        self.M, self.M_local = keys.shape[0], n_own

    def keys(self):
        return self._keys

    def neighbors(self):
        return self._nbr

    def splat(self, src):
        s = src.cpu().numpy().astype(np.float32)
        vert = np.zeros((self.M, s.shape[1]), np.float32)
        for r in range(self.d + 1):
            np.add.at(vert, self._vid[:, r], self._w[:, r:r + 1] * s)
        return torch.from_numpy(vert)

    def blur(self, vert):
        v = vert.cpu().numpy().astype(np.float32)
        for a in range(self.d + 1):
            n1, n2 = self._nbr[a, :, 0], self._nbr[a, :, 1]
            z = np.zeros((1, v.shape[1]), np.float32)
            ve = np.concatenate([v, z])                  # row -1 = zeros (absent neighbour, :516-522)
            v = (np.float32(2) * (np.float32(0.25) * ve[n1] + np.float32(0.5) * v + np.float32(0.25) * ve[n2])).astype(np.float32)
        return torch.from_numpy(v)

    def slice(self, vert, out=None):
        v = vert.cpu().numpy().astype(np.float32)
        c = np.float32(1.0 + 2.0 ** -self.d)
        res = np.zeros((self.n, v.shape[1]), np.float32)
        for r in range(self.d + 1):
            res += (self._w[:, r:r + 1] * v[self._vid[:, r]]) / c
        res = torch.from_numpy(res)
        if out is not None:
            out.copy_(res)
            return out
        return res


class _WholeOracle:
    """The whole image's lattice with the reference's table behaviour (faithful CPU oracle), offering what
    rowtile.RowBand(table="reference") asks of an engine: keys, vertices_of_pixels, sub_lattice."""

    def __init__(self, ref):
        self._o = po.Oracle(np.ascontiguousarray(ref.cpu().numpy(), np.float32), faithful_table=True)
        self.M = self._o.M
        self._keys = self._o.keys()
        self._vid, self._w = self._o.replay()
        self._nbr = np.asarray(self._o.neighbors())

    def keys(self):
        return self._keys

    def vertices_of_pixels(self, p0, p1):
        m = np.zeros(self.M, bool)
        m[self._vid[p0:p1].ravel()] = True
        return m

    def sub_lattice(self, p0, p1, sel, n_own, ref_band):
        sel = np.asarray(sel, np.int64)
        pos = np.full(self.M, -1, np.int64)
        pos[sel] = np.arange(len(sel))
        vid = pos[self._vid[p0:p1]]
        assert vid.min() >= 0 and vid.max() < n_own
        nb = self._nbr[:, sel, :]
        nbr = np.where(nb >= 0, pos[np.clip(nb, 0, None)], -1)
        return _NumpyBand(self._keys[sel], vid, self._w[p0:p1], nbr, n_own)


class OracleEngineRef(OracleEngine):
    """OracleEngine that can also cut bands out of the whole image's reference-table lattice (RowBand table="reference")."""

    @classmethod
    def whole_image(cls, ref):
        return _WholeOracle(ref)

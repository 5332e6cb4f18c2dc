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

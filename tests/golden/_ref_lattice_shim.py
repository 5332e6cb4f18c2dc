"""Container-only helper for tests/golden/generate.py: stands where the reference's JIT-built
``lattice`` extension would (crf/gaussian_matrix.py:15-16) and forwards ``filter(src, ref)`` to
the reference's own C++ engine built by oracle/build_ref.sh.  A real module (not a lambda) so
that the reference's mp.Pool in batched_filter (gaussian_matrix.py:370-377) can pickle it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import phl_oracle as _po  # noqa: E402


def filter(src, ref):
    assert src.dtype == torch.float32 and ref.dtype == torch.float32  # accessor<float,2>, permutohedral.h:214
    out = _po.reference_filter(src.detach().numpy(), ref.detach().numpy())
    return torch.from_numpy(np.ascontiguousarray(out))

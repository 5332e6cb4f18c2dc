"""Legacy module name: Experiments/DenseCrf.ipynb cell 2 does ``from crf.crf import *`` and then
uses the mean-field helpers; the cost-volume helpers come from ``crf.depth`` in the same cell."""
from crf.crf_module import *  # noqa: F401,F403

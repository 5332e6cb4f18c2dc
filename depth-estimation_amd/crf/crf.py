"""Legacy module name: Experiments/DenseCrf.ipynb cell 2 does ``from crf.crf import *``."""
from crf.crf_module import *  # noqa: F401,F403

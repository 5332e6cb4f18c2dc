"""``crf`` -- the reference's package name (mfinzi/depth-estimation, crf/), so that
``from crf.crf_module import mean_field_infer`` and ``from crf.gaussian_matrix import
LatticeGaussian`` keep working.  Only the dense-CRF mean-field path and its lattice operators
live here; the lattice filter itself is the HIP library behind ``phl`` / ``lattice``."""

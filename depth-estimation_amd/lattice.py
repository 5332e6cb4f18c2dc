"""``lattice`` -- same module name and single entry point as the reference's JIT-built pybind
extension (crf/lattice/lite/lattice.cpp:14-15: ``m.def("filter", &filter, "lattice filter")``),
backed by the HIP kernels through the C ABI (include/phl.h)."""
from phl import filter  # noqa: F401

// lattice_ext.cpp -- the reference's L1 layer, re-done over the C ABI.
//
// The reference binds its engine with a 15-line pybind extension
//     at::Tensor filter(at::Tensor src, at::Tensor ref)      crf/lattice/lite/lattice.cpp:6-15
// This is the same function over include/phl.h: tensors in, tensor out, no kernels here.
// It exists to show the boundary from the C++ side; the Python
// package uses the ctypes binding in phl/__init__.py, which adds the lattice cache.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "phl.h"

at::Tensor filter(at::Tensor src, at::Tensor ref)
{
    TORCH_CHECK(src.dim() == 2 && ref.dim() == 2 && src.size(0) == ref.size(0), "Incompatible shapes ", src.sizes(), ", and ",
                ref.sizes());
    TORCH_CHECK(src.scalar_type() == at::kFloat && ref.scalar_type() == at::kFloat,
                "lattice.filter is float32 only (as the reference: permutohedral.h:214-215)");
    const auto in_device = src.device();
    TORCH_CHECK(phl_device_count() > 0, "no HIP device: the lattice filter has no CPU fallback");
    const int dev = src.is_cuda() ? src.get_device() : c10::hip::current_device();
    const auto gpu = at::Device(at::kCUDA, dev);
    at::Tensor s = src.to(gpu), r = ref.to(gpu);
    at::Tensor out = at::empty({s.size(0), s.size(1)}, s.options());
    auto stream = c10::hip::getCurrentHIPStream(dev);
    const int rc = phl_filter_once(s.data_ptr<float>(), (int)s.size(1), s.stride(0), s.stride(1), r.data_ptr<float>(),
                                   (int)r.size(1), r.stride(0), r.stride(1), s.size(0), out.data_ptr<float>(), out.stride(0),
                                   out.stride(1), PHL_FILTER_DEFAULT, dev, (phl_stream)stream.stream());
    TORCH_CHECK(rc == PHL_OK, "phl error ", rc, ": ", phl_last_error());
    return out.to(in_device);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("filter", &filter, "lattice filter"); }

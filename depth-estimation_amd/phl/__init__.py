"""phl -- Python binding of the MI355X permutohedral-lattice filter (C ABI in include/phl.h).

Thin plumbing only: torch supplies device memory and the current HIP stream, ctypes calls
``lib/libphl.so``.  There is NO CPU implementation behind this module: if the shared library or
a HIP device is missing, calls raise ``RuntimeError``.  CPU tensors are accepted the way the
reference accepts them (its extension is CPU-only, crf/lattice/lite/permutohedral.h:214) but
are computed on the GPU and copied back.

Public surface
    Lattice(ref)                 build once per feature tensor  (init-once / filter-many)
    Lattice.filter(src, ...)     == reference ``lattice.filter(src, ref)`` for that ref
    filter(src, ref)             drop-in for ``latticefilter`` (crf/gaussian_matrix.py:15-16);
                                 lattices are cached per ``ref`` tensor, invisibly
"""
import ctypes as C
import os
import threading
from collections import OrderedDict

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHL_LIB: another build of the same library (A/B timing of two source states on one box); never a fallback
LIB_PATH = os.environ.get("PHL_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libphl.so")

SUBTRACT_INPUT = 1
EXACT = 4
NO_TILES = 8
BUILD_REFERENCE_TABLE = 1

_f32p = C.c_void_p
_lib = None
_lib_lock = threading.Lock()


class PhlError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"phl error {status}: {message}")
        self.status = status


def load_library():
    """Load lib/libphl.so (built by ``python __graft_entry__.py`` / ``make -C csrc``)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py`). "
                "There is no CPU fallback for the lattice filter.")
        lib = C.CDLL(LIB_PATH)
        i64, i32, vp, u32 = C.c_int64, C.c_int, C.c_void_p, C.c_uint
        lib.phl_version.restype = i32
        lib.phl_last_error.restype = C.c_char_p
        lib.phl_status_string.restype = C.c_char_p
        lib.phl_status_string.argtypes = [i32]
        lib.phl_device_count.restype = i32
        lib.phl_build.argtypes = [C.POINTER(vp), vp, i64, i32, i64, i64, i32, vp]
        lib.phl_build_ex.argtypes = [C.POINTER(vp), vp, i64, i32, i64, i64, i32, vp, u32]
        lib.phl_debug_reference_table.argtypes = [vp, vp, i64, i32, i64, vp, i64, C.POINTER(i64), vp, vp, i32,
                                                  C.POINTER(i32), C.POINTER(i32)]
        if hasattr(lib, "phl_debug_probe_paths"):       # (test hook; an older build loaded through PHL_LIB lacks it)
            lib.phl_debug_probe_paths.argtypes = [vp, i64, i32, vp, i32, vp, i32, C.c_uint64, vp, i32, i32, C.POINTER(i32)]
        lib.phl_destroy.argtypes = [vp]
        for name in ("phl_num_pixels", "phl_num_vertices", "phl_device_bytes"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [vp]
        lib.phl_num_dims.argtypes = [vp]
        lib.phl_device.argtypes = [vp]
        lib.phl_filter_grad.argtypes = [vp, vp, i64, vp, i64, i32, vp, i64, i64, vp, vp, i64, vp]
        lib.phl_reserve.argtypes = [vp, i32]
        lib.phl_reserve_ex.argtypes = [vp, i32, u32]
        lib.phl_add_vertices.argtypes = [vp, vp, i64, vp, vp]
        lib.phl_num_local_vertices.restype = i64
        lib.phl_num_local_vertices.argtypes = [vp]
        lib.phl_filter.argtypes = [vp, vp, i32, i64, i64, vp, i64, i64, u32, vp]
        lib.phl_filter_once.argtypes = [vp, i32, i64, i64, vp, i32, i64, i64, i64, vp, i64, i64, u32, i32, vp]
        lib.phl_splat.argtypes = [vp, vp, i32, i64, vp, u32, vp]
        lib.phl_tile_stats.argtypes = [vp, i32, vp]
        for name in ("phl_num_chunks", "phl_partial_rows"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [vp]
        lib.phl_chunks_touching.argtypes = [vp, vp, i64, vp, vp]
        lib.phl_splat_part.argtypes = [vp, vp, i32, i64, vp, vp, vp, i64, vp, i64, vp]
        if hasattr(lib, "phl_splat_part_pack"):         # (an older build loaded through PHL_LIB lacks these)
            lib.phl_splat_part_pack.argtypes = [vp, vp, i32, i64, vp, vp, vp, i64, vp, i64, vp, vp, i64, vp]
            lib.phl_set_blur_rows.argtypes = [vp, vp, i32]
        lib.phl_blur_axis.argtypes = [vp, i32, vp, vp, i32, vp]
        lib.phl_blur.argtypes = [vp, vp, vp, i32, C.POINTER(i32), vp]
        lib.phl_gather_rows.argtypes = [vp, i32, vp, i64, vp, i64, vp]
        lib.phl_scatter_add_rows.argtypes = [vp, i32, vp, i64, vp, i64, vp]
        lib.phl_slice.argtypes = [vp, vp, i32, vp, i64, vp, i64, u32, vp]
        lib.phl_softmax_neg_add.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, vp]
        lib.phl_expected_value.argtypes = [vp, i64, vp, vp, i64, i32, vp]
        lib.phl_compat_softmax.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i32, u32, vp]
        lib.phl_uniform_compat_softmax.argtypes = [vp, i64, vp, i64, C.c_float, C.c_float, vp, i64, i64, i32, u32, vp]
        if hasattr(lib, "phl_compat_softmax_split"):
            lib.phl_compat_planes_bytes.argtypes = [i32]
            lib.phl_compat_planes_bytes.restype = C.c_size_t
            lib.phl_compat_prepare.argtypes = [vp, i32, vp, vp]
            lib.phl_compat_softmax_split.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, u32, vp]
        lib.phl_stream_copy.argtypes = [vp, vp, i64, vp]
        lib.phl_copy2d.argtypes = [vp, i64, i64, vp, i64, i64, i64, i32, vp]
        lib.phl_cost_volume.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp]
        lib.phl_get_keys.argtypes = [vp, vp]
        lib.phl_get_vertex_order.argtypes = [vp, vp]
        lib.phl_get_replay.argtypes = [vp, vp, vp]
        lib.phl_get_neighbors.argtypes = [vp, vp]
        lib.phl_get_splat_lists.argtypes = [vp, vp, vp, vp]
        _lib = lib
        return lib


def _check(rc):
    if rc != 0:
        raise PhlError(rc, load_library().phl_last_error().decode("utf-8", "replace"))


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("phl: no HIP device visible to torch; the lattice filter has no CPU fallback")


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_STAGE_BYTES = 2 << 20       # piece size of the staged form below


def to_device(t, device):
    """``t.to(device)`` for the reference's calling convention -- pageable CPU tensors (DenseCrf.ipynb:142-152 hands
    CPU tensors to mean_field_infer).  Enqueued on the current stream.

    The runtime's own pageable path is the fast one on this platform (tools/h2d_probe.py: 44-52 GB/s from pageable memory
    at 7 MB ... 1 GB -- it pins the user's pages for the DMA); copying through pinned staging pieces first, as rounds 3-4 did,
    is bound by the host memcpy and by the caching host allocator handing out fresh pinned blocks while earlier pieces are
    still in flight (3-4 GB/s at 128 MB and up, 20 GB/s at 7 MB on the same box; box to box the notebook-sized call swung
    between 1.2 and 5 ms).  PHL_H2D=staged brings the staged form back."""
    device = torch.device(device)
    if t.device == device:
        return t
    if (os.environ.get("PHL_H2D") != "staged" or t.is_cuda or t.is_pinned() or not t.is_contiguous()
            or t.numel() * t.element_size() < (1 << 20)):
        return t.to(device, non_blocking=(not t.is_cuda and t.is_pinned()))
    out = torch.empty(t.shape, dtype=t.dtype, device=device)
    src, dst = t.reshape(-1), out.view(-1)
    step = max(1, _STAGE_BYTES // t.element_size())
    for a in range(0, src.numel(), step):
        b = min(src.numel(), a + step)
        stage = torch.empty(b - a, dtype=t.dtype, pin_memory=True)
        stage.copy_(src[a:b])
        dst[a:b].copy_(stage, non_blocking=True)     # (the host allocator keeps the piece until this copy has run)
    return out


def to_host(t):
    """Device tensor -> CPU tensor in pinned memory (one DMA, no bounce through the runtime's staging buffers);
    returns after the copy has completed."""
    if not t.is_cuda:
        return t
    if t.numel() * t.element_size() < (1 << 20):
        return t.cpu()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return out


def _as_device(t, device):
    if t.dtype != torch.float32:
        # the reference extension is float-only (accessor<float,2>, permutohedral.h:214-215)
        raise TypeError(f"phl: expected float32 tensor, got {t.dtype}")
    return t if t.device == device else to_device(t, device)


_parked = []    # lattice handles whose owner died during a stream capture; freed by the next close()


class Lattice:
    """Permutohedral lattice of a feature tensor ``ref`` [n, d] (fp32, any strides).

    Build cost is paid once; ``filter`` then runs splat -> blur -> slice for any [n, vd] values.
    Re-entrant: any number of threads / streams may filter through one Lattice at the same time (the C library
    hands every call its own value workspace; see phl_reserve in include/phl.h).
    """

    def __init__(self, ref, device=None, reference_table=False):
        """reference_table=True: reproduce the reference's hash-table behaviour across its doublings (duplicate
        vertices above M = 16383, see PHL_BUILD_REFERENCE_TABLE in include/phl.h) -- with ``exact=True`` the
        filter is then bit-identical to the reference's CPU path at any size."""
        _require_gpu()
        lib = load_library()
        if ref.dim() != 2:
            raise ValueError(f"ref must be [n, d], got {tuple(ref.shape)}")
        if device is None:
            device = ref.device if ref.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        ref_d = _as_device(ref.detach(), self.device)
        self.n, self.d = int(ref_d.shape[0]), int(ref_d.shape[1])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.phl_build_ex(C.byref(handle), C.c_void_p(ref_d.data_ptr()), self.n, self.d, ref_d.stride(0),
                                    ref_d.stride(1), self.device.index or 0, _stream(self.device),
                                    BUILD_REFERENCE_TABLE if reference_table else 0))
        self.reference_table = bool(reference_table)
        self._h = handle
        self.M = int(lib.phl_num_vertices(handle))

    # ---- a band cut out of the whole image's lattice (row-band multi-GPU, phl/rowtile.py) ---------------------
    @classmethod
    def whole_image(cls, ref, device=None):
        """The lattice row bands are cut from: the whole image's, with the reference's table behaviour."""
        return cls(ref, device=device, reference_table=True)

    def vertices_of_pixels(self, p0, p1):
        """bool numpy [M]: first-touch vertices touched by pixels [p0, p1)."""
        lib = load_library()
        lib.phl_vertices_of_pixels.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        mask = np.zeros(self.M, np.uint8)
        with torch.cuda.device(self.device):
            _check(lib.phl_vertices_of_pixels(self._h, int(p0), int(p1), mask.ctypes.data_as(C.c_void_p), _stream(self.device)))
        return mask.astype(bool)

    def sub_lattice(self, p0, p1, sel, n_own, ref_band):
        """Lattice of pixels [p0, p1) on the selected vertices ``sel`` (first-touch ids; the first n_own = every vertex those
        pixels touch, then ghosts in the caller's order): phl_sub_lattice in include/phl.h.  The new lattice numbers its
        vertices by position in ``sel``; rows of ghost vertices are M_own + position among the ghosts."""
        lib = load_library()
        lib.phl_sub_lattice.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        sel = np.ascontiguousarray(sel, np.int32)
        ref_d = _as_device(ref_band.detach(), self.device)
        assert ref_d.shape == (p1 - p0, self.d)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.phl_sub_lattice(C.byref(handle), self._h, int(p0), int(p1), sel.ctypes.data_as(C.c_void_p), len(sel), int(n_own),
                                       C.c_void_p(ref_d.data_ptr()), ref_d.stride(0), ref_d.stride(1), _stream(self.device)))
        sub = Lattice.__new__(Lattice)
        sub.device, sub.n, sub.d = self.device, int(p1 - p0), self.d
        sub.reference_table = self.reference_table
        sub._h = handle
        sub.M = int(lib.phl_num_vertices(handle))
        return sub

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:          # at interpreter shutdown the module globals may be gone
            try:
                # phl_destroy hipFree()s; inside a stream capture that would invalidate the capture
                # (a garbage-collected Lattice can land here at any time), so park the handle instead
                if torch.cuda.is_current_stream_capturing():
                    _parked.append(h)
                    return
                _lib.phl_destroy(h)
                while _parked:
                    _lib.phl_destroy(_parked.pop())
            except Exception:
                pass

    __del__ = close

    @property
    def device_bytes(self):
        return int(load_library().phl_device_bytes(self._h))

    def reserve(self, vd, strided_io=False, exact=False):
        """Pre-size everything a ``filter`` over ``vd`` channels needs, so that the next call -- on any stream,
        e.g. inside a HIP-graph capture -- allocates nothing.  strided_io: also the staging copies channel-major
        (NCHW) views -- and, from 128 channels on, pixel-major rows off the 16-byte grid such as column slices -- go
        through; exact: prepare for ``exact=True`` calls."""
        _check(load_library().phl_reserve_ex(self._h, int(vd), (1 if strided_io else 0) | (2 if exact else 0)))

    # ---- hot path ---------------------------------------------------------------------------
    def filter(self, src, subtract_input=False, out=None, exact=False, no_tiles=False):
        """``lattice.filter(src, ref)`` of the reference; result lives on ``src``'s device.

        exact: reference-exact arithmetic (pixel-ordered splat sums, per-term divide in slice):
        bit-identical to the reference's CPU path.  The default LDS-staged path agrees to fp32
        rounding (~1e-7 relative).  no_tiles: plain gather kernels (A/B)."""
        if src.dim() != 2 or src.shape[0] != self.n:
            # same text as the reference's assert (gaussian_matrix.py:429-430)
            raise AssertionError("Incompatible shapes {}, and {}".format(tuple(src.shape), (self.n, self.d)))
        src_d = _as_device(src.detach(), self.device)
        vd = int(src_d.shape[1])
        res = out if (out is not None and out.device == self.device) else torch.empty(
            (self.n, vd), dtype=torch.float32, device=self.device)
        flags = (SUBTRACT_INPUT if subtract_input else 0) | (EXACT if exact else 0) | (NO_TILES if no_tiles else 0)
        with torch.cuda.device(self.device):
            _check(load_library().phl_filter(self._h, C.c_void_p(src_d.data_ptr()), vd, src_d.stride(0),
                                             src_d.stride(1), C.c_void_p(res.data_ptr()), res.stride(0),
                                             res.stride(1), flags, _stream(self.device)))
        if out is not None and out is not res:
            out.copy_(res)
            return out
        if src.device == self.device:
            return res
        return to_host(res) if src.device.type == "cpu" else res.to(src.device)

    # ---- stages (profiling / parity of intermediates) ---------------------------------------
    def splat(self, src, exact=False, no_tiles=False, out=None):
        src_d = _as_device(src.detach(), self.device)
        assert src_d.stride(1) == 1, "stage API takes pixel-major rows"
        vd = int(src_d.shape[1])
        vert = torch.empty((self.M, vd), dtype=torch.float32, device=self.device) if out is None else out
        assert vert.shape == (self.M, vd) and vert.is_contiguous()
        with torch.cuda.device(self.device):
            _check(load_library().phl_splat(self._h, C.c_void_p(src_d.data_ptr()), vd, src_d.stride(0),
                                            C.c_void_p(vert.data_ptr()),
                                            (EXACT if exact else 0) | (NO_TILES if no_tiles else 0),
                                            _stream(self.device)))
        return vert

    # ---- splat in parts (row-band exchange overlap, see phl_splat_part in include/phl.h) ------------------
    def chunks_touching(self, rows):
        """bool numpy mask [#chunks]: pixel chunks that contribute to any of the vertex rows (int64 device tensor)."""
        lib = load_library()
        mask = np.zeros(int(lib.phl_num_chunks(self._h)), np.int32)
        rows = rows.to(self.device, torch.int64).contiguous()
        with torch.cuda.device(self.device):
            _check(lib.phl_chunks_touching(self._h, C.c_void_p(rows.data_ptr()), int(rows.numel()),
                                           mask.ctypes.data_as(C.c_void_p), _stream(self.device)))
        return mask.astype(bool)

    @property
    def partial_rows(self):
        return int(load_library().phl_partial_rows(self._h))

    def splat_part(self, src, out, partial, chunks, rows, pack_pos=None, pack=None):
        """Run the chunk splat for the listed chunks (int32 device tensor) and complete the listed vertex rows
        (int32 device tensor) of ``out`` [M, vd]; ``partial`` [partial_rows, vd] is shared by the parts of one splat.
        pack_pos (int32 device tensor, one per listed row) / pack [*, vd]: listed row i is also written to
        pack[pack_pos[i]] by the kernel that completes it (the row-band exchange's send buffer)."""
        vd = int(src.shape[1])
        assert src.stride(1) == 1 and out.is_contiguous() and out.shape == (self.M, vd)
        assert chunks.dtype == torch.int32 and rows.dtype == torch.int32 and chunks.is_contiguous() and rows.is_contiguous()
        assert partial.is_contiguous() and partial.shape[0] >= self.partial_rows and partial.shape[1] == vd
        lib = load_library()
        with torch.cuda.device(self.device):
            if pack_pos is None:
                _check(lib.phl_splat_part(self._h, C.c_void_p(src.data_ptr()), vd, src.stride(0), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(partial.data_ptr()), C.c_void_p(chunks.data_ptr()), int(chunks.numel()),
                                          C.c_void_p(rows.data_ptr()), int(rows.numel()), _stream(self.device)))
            else:
                assert pack_pos.dtype == torch.int32 and pack_pos.is_contiguous() and pack_pos.numel() == rows.numel()
                assert pack.stride(1) == 1 and pack.shape[1] == vd and pack.dtype == torch.float32
                _check(lib.phl_splat_part_pack(self._h, C.c_void_p(src.data_ptr()), vd, src.stride(0), C.c_void_p(out.data_ptr()),
                                               C.c_void_p(partial.data_ptr()), C.c_void_p(chunks.data_ptr()), int(chunks.numel()),
                                               C.c_void_p(rows.data_ptr()), int(rows.numel()), C.c_void_p(pack_pos.data_ptr()),
                                               C.c_void_p(pack.data_ptr()), pack.stride(0), _stream(self.device)))
        return out

    def set_blur_rows(self, ranges):
        """Row-band lattices: per blur axis the rows whose output of that axis is read later, as three ascending
        {begin, end} row ranges (numpy / list [d+1][3][2]); None: all rows.  See phl_set_blur_rows in include/phl.h."""
        lib = load_library()
        if ranges is None:
            _check(lib.phl_set_blur_rows(self._h, None, 0))
            return
        r = np.ascontiguousarray(ranges, np.int64)
        assert r.shape == (self.d + 1, 3, 2), r.shape
        _check(lib.phl_set_blur_rows(self._h, r.ctypes.data_as(C.c_void_p), self.d + 1))

    def blur_axis(self, axis, vin, vout=None):
        vd = int(vin.shape[1])
        vout = torch.empty_like(vin) if vout is None else vout
        assert vin.is_contiguous() and vout.is_contiguous()
        with torch.cuda.device(self.device):
            _check(load_library().phl_blur_axis(self._h, int(axis), C.c_void_p(vin.data_ptr()),
                                                C.c_void_p(vout.data_ptr()), vd, _stream(self.device)))
        return vout

    def blur(self, vert, scratch=None):
        """All d+1 axes (two per pass).  ``vert`` is overwritten (it is one of the two ping-pong buffers)."""
        other = torch.empty_like(vert) if scratch is None else scratch
        assert vert.is_contiguous() and other.is_contiguous() and other.shape == vert.shape
        which = C.c_int(0)
        with torch.cuda.device(self.device):
            _check(load_library().phl_blur(self._h, C.c_void_p(vert.data_ptr()), C.c_void_p(other.data_ptr()),
                                           int(vert.shape[1]), C.byref(which), _stream(self.device)))
        return other if which.value else vert

    def gather_rows(self, vert, idx, out=None):
        """out[r] = vert[idx[r]] (idx: int64 device tensor).  Row-band exchange helper."""
        k, vd = int(idx.numel()), int(vert.shape[1])
        out = torch.empty((k, vd), dtype=torch.float32, device=self.device) if out is None else out
        assert vert.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and out.stride(1) == 1
        with torch.cuda.device(self.device):
            _check(load_library().phl_gather_rows(C.c_void_p(vert.data_ptr()), vd, C.c_void_p(idx.data_ptr()), k,
                                                  C.c_void_p(out.data_ptr()), out.stride(0) if k else vd, _stream(self.device)))
        return out

    def scatter_add_rows(self, vert, idx, rows):
        """vert[idx[r]] += rows[r]; idx must hold distinct vertex ids."""
        k, vd = int(idx.numel()), int(vert.shape[1])
        assert vert.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and (k == 0 or rows.stride(1) == 1)
        with torch.cuda.device(self.device):
            _check(load_library().phl_scatter_add_rows(C.c_void_p(vert.data_ptr()), vd, C.c_void_p(idx.data_ptr()), k,
                                                       C.c_void_p(rows.data_ptr()), rows.stride(0) if k else vd,
                                                       _stream(self.device)))
        return vert

    def slice(self, vert, sub=None, exact=False, out=None, no_tiles=False):
        vd = int(vert.shape[1])
        out = torch.empty((self.n, vd), dtype=torch.float32, device=self.device) if out is None else out
        with torch.cuda.device(self.device):
            _check(load_library().phl_slice(self._h, C.c_void_p(vert.data_ptr()), vd, C.c_void_p(out.data_ptr()),
                                            out.stride(0), C.c_void_p(sub.data_ptr()) if sub is not None else None,
                                            sub.stride(0) if sub is not None else 0,
                                            (EXACT if exact else 0) | (NO_TILES if no_tiles else 0),
                                            _stream(self.device)))
        return out

    def filter_grad(self, src, g, ref, need_src=True):
        """Gradients of ``sum(g * filter(src, ref))``: returns (grad_src or None, grad_ref [n, d]) -- the body of
        LatticeFilter.backward (crf/gaussian_matrix.py:435-468) in two fused passes (phl_filter_grad in
        include/phl.h).  Raises PhlError(status UNSUPPORTED) for shapes the fused path does not take."""
        src_d = _as_device(src.detach(), self.device).contiguous()
        g_d = _as_device(g.detach(), self.device).contiguous()
        ref_d = _as_device(ref.detach(), self.device)
        assert src_d.shape == g_d.shape and src_d.shape[0] == self.n and ref_d.shape == (self.n, self.d)
        L = int(src_d.shape[1])
        grad_ref = torch.empty((self.n, self.d), dtype=torch.float32, device=self.device)
        grad_src = torch.empty((self.n, L), dtype=torch.float32, device=self.device) if need_src else None
        with torch.cuda.device(self.device):
            _check(load_library().phl_filter_grad(self._h, C.c_void_p(src_d.data_ptr()), src_d.stride(0),
                                                  C.c_void_p(g_d.data_ptr()), g_d.stride(0), L,
                                                  C.c_void_p(ref_d.data_ptr()), ref_d.stride(0), ref_d.stride(1),
                                                  C.c_void_p(grad_ref.data_ptr()),
                                                  C.c_void_p(grad_src.data_ptr()) if need_src else None, L,
                                                  _stream(self.device)))
        return grad_src, grad_ref

    def tile_stats(self, vd):
        """Chunk statistics of the LDS-staged path (see phl_tile_stats in include/phl.h)."""
        out = np.zeros(7, np.int64)
        _check(load_library().phl_tile_stats(self._h, int(vd), out.ctypes.data_as(C.c_void_p)))
        keys = ("pixels_per_chunk", "chunks", "max_local_vertices", "slots", "multi_chunk_slots", "staged_splat",
                "staged_slice")
        return dict(zip(keys, (int(x) for x in out)))

    def add_vertices(self, keys):
        """Row-band support: append neighbouring-band vertices (distinct int16 keys [K, d]) as
        ghosts; returns, for every key, its ROW in the vertex buffers (int32 [K]).  Updates ``M``."""
        keys = np.ascontiguousarray(keys, np.int16).reshape(-1, self.d)
        vid = np.empty(len(keys), np.int32)
        with torch.cuda.device(self.device):
            _check(load_library().phl_add_vertices(self._h, keys.ctypes.data_as(C.c_void_p), len(keys),
                                                   vid.ctypes.data_as(C.c_void_p), _stream(self.device)))
        self.M = int(load_library().phl_num_vertices(self._h))
        self._rows = None
        return vid

    @property
    def M_local(self):
        return int(load_library().phl_num_local_vertices(self._h))

    # ---- vertex numbering ---------------------------------------------------------------------
    # keys() / replay() / neighbors() / splat_lists() speak the reference's first-touch vertex ids.  The ROWS of
    # the [M, vd] buffers splat / blur / slice exchange are in the library's internal (locality) order.
    def vertex_rows(self):
        """int64 device tensor [M]: row of first-touch vertex v in the vertex buffers (cached)."""
        hit = getattr(self, "_rows", None)
        if hit is None or hit.numel() != self.M:
            out = np.empty(self.M, np.int32)
            _check(load_library().phl_get_vertex_order(self._h, out.ctypes.data_as(C.c_void_p)))
            hit = self._rows = torch.from_numpy(out.astype(np.int64)).to(self.device)
        return hit

    def to_first_touch(self, vert):
        """Vertex buffer with its rows re-ordered to first-touch vertex order (parity checks against the CPU path)."""
        return vert.index_select(0, self.vertex_rows())

    def from_first_touch(self, vert_ft):
        """Inverse of to_first_touch: rows in first-touch order -> a buffer in internal row order."""
        out = torch.empty_like(vert_ft)
        out.index_copy_(0, self.vertex_rows(), vert_ft)
        return out

    # ---- introspection (host copies) --------------------------------------------------------
    def pixel_order(self):
        """int32 [n]: pixels in chunk order (chunk c = pixel_order()[c*P:(c+1)*P], P = tile_stats()['pixels_per_chunk'])."""
        out = np.empty(self.n, np.int32)
        lib = load_library()
        lib.phl_get_pixel_order.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib.phl_get_pixel_order(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def keys(self):
        out = np.empty((self.M, self.d), np.int16)
        _check(load_library().phl_get_keys(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def replay(self):
        vid = np.empty((self.n, self.d + 1), np.int32)
        w = np.empty((self.n, self.d + 1), np.float32)
        _check(load_library().phl_get_replay(self._h, vid.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)))
        return vid, w

    def neighbors(self):
        out = np.empty((self.d + 1, self.M, 2), np.int32)
        _check(load_library().phl_get_neighbors(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def splat_lists(self):
        ptr = np.empty(self.M + 1, np.int32)
        pix = np.empty(self.n * (self.d + 1), np.int32)
        w = np.empty(self.n * (self.d + 1), np.float32)
        _check(load_library().phl_get_splat_lists(self._h, ptr.ctypes.data_as(C.c_void_p),
                                                  pix.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)))
        return ptr, pix, w


# ---------------------------------------------------------------------------------------------
# fused elementwise steps of the mean-field iteration (crf/crf_module.py:49-52)
def _rowmajor(t):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1


def softmax_neg_add(E0, G=None, out=None):
    """softmax(-(E0 + G), dim=1) in one pass over HBM (G optional).  CUDA fp32 [n, L] only."""
    if not _rowmajor(E0) or (G is not None and (not _rowmajor(G) or G.shape != E0.shape)):
        raise TypeError("softmax_neg_add: expects fp32 CUDA [n, L] tensors with unit channel stride")
    n, L = E0.shape
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=E0.device)
    with torch.cuda.device(E0.device):
        _check(load_library().phl_softmax_neg_add(
            C.c_void_p(E0.data_ptr()), E0.stride(0), C.c_void_p(G.data_ptr()) if G is not None else None,
            G.stride(0) if G is not None else 0, C.c_void_p(out.data_ptr()), out.stride(0), n, L, _stream(E0.device)))
    return out


_mu_t_cache = {}


def _mu_transposed(Mu, device):
    """Mu^T as a dense fp32 device tensor padded to the MFMA tile width, cached per (storage, version): Mu is
    fixed during inference."""
    key = (Mu.data_ptr(), Mu._version, tuple(Mu.shape), tuple(Mu.stride()), str(device))
    hit = _mu_t_cache.get(key)
    if hit is None:
        if len(_mu_t_cache) > 8:
            _mu_t_cache.clear()
        L = Mu.shape[0]
        Lp = (L + 31) // 32 * 32                 # the kernel's tile width: zero padding beyond L
        mt = torch.zeros((Lp, Lp), dtype=torch.float32, device=device)
        mt[:L, :L] = Mu.detach().to(device, torch.float32).t()
        if not torch.cuda.is_current_stream_capturing():     # cached across calls: complete before another stream can use it
            torch.cuda.current_stream(device).synchronize()
        hit = _mu_t_cache[key] = (mt, Mu)        # keeps Mu alive
    return hit[0]


_mu_planes_cache = {}


def _mu_planes(Mu, mu_t, device):
    """The compatibility matrix as the split kernel reads it (phl_compat_prepare), cached like Mu^T."""
    key = (Mu.data_ptr(), Mu._version, tuple(Mu.shape), tuple(Mu.stride()), str(device))
    hit = _mu_planes_cache.get(key)
    if hit is None:
        if len(_mu_planes_cache) > 8:
            _mu_planes_cache.clear()
        lib = load_library()
        planes = torch.empty((lib.phl_compat_planes_bytes(Mu.shape[0]),), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _check(lib.phl_compat_prepare(C.c_void_p(mu_t.data_ptr()), Mu.shape[0], C.c_void_p(planes.data_ptr()), _stream(device)))
            # cached across calls, and a later call may come on another stream: finished before anyone else can see it
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(device).synchronize()
        hit = _mu_planes_cache[key] = (planes, Mu, mu_t)
    return hit[0]


_mu_uniform_cache = {}


def _mu_uniform(Mu):
    """(alpha, beta) if Mu == alpha * ones + beta * eye exactly (the Potts family: the reference's ``potts`` layer is
    (1, -1), crf_module.py:55-64), else None.  One device -> host read per (storage, version) of Mu."""
    key = (Mu.data_ptr(), Mu._version, tuple(Mu.shape), tuple(Mu.stride()))
    if key in _mu_uniform_cache:
        return _mu_uniform_cache[key][0]
    if len(_mu_uniform_cache) > 8:
        _mu_uniform_cache.clear()
    L = Mu.shape[0]
    res = None
    m = Mu.detach().to(torch.float32)
    if L >= 2:
        alpha = m[0, 1]
        beta = m[0, 0] - alpha
        if bool(((m - alpha) - beta * torch.eye(L, dtype=torch.float32, device=m.device) == 0).all()) and bool(torch.isfinite(m).all()):
            res = (float(alpha), float(beta))
    _mu_uniform_cache[key] = (res, Mu)           # keeps Mu alive
    return res


def compat_softmax(E0, X, Mu, out=None, logits=False, structure=True, arith=None, uniform=None):
    """softmax(-(E0 + X @ Mu), dim=1): the whole non-lattice half of a mean-field iteration
    (crf/crf_module.py:51-52) for fp32 CUDA E0, X [n, L] and Mu [L, L], in one fused MFMA kernel
    (phl_compat_softmax) when L % 4 == 0 and L <= 256 (label counts that are not a multiple of 32 run on a padded
    tile); other label counts take a library GEMM followed by the fused add + softmax pass.  A Mu of the Potts family
    (alpha * ones + beta * eye, detected once per Mu; ``structure=False`` switches that off) needs no product at all:
    phl_uniform_compat_softmax streams E0, X and Q once.  logits=True returns -(E0 + X @ Mu) instead (CRFasRNN's output).
    arith: "f32" = the f32-input matrix cores (bitwise an fma chain in k order), "split" = bf16 matrix cores on operands
    split three ways, six partial products, f32 accumulation (phl_compat_softmax_split: a 256-label tile, 128 < L <= 256;
    the same accuracy against float64, 3/8 of the matrix time).  Default: "split" for L > 176 -- below that the f32 kernel
    is bound by its bytes as well and computes no padding -- unless PHL_COMPAT_ARITH names one of the two."""
    if arith is None:
        arith = os.environ.get("PHL_COMPAT_ARITH") or ("split" if E0.shape[-1] > 176 else "f32")
    if arith not in ("f32", "split"):
        raise ValueError(f"compat_softmax: arith must be 'f32' or 'split', got {arith!r}")
    if not (_rowmajor(E0) and _rowmajor(X) and X.shape == E0.shape and Mu.shape == (E0.shape[1], E0.shape[1])):
        raise TypeError("compat_softmax: expects fp32 CUDA E0, X [n, L] with unit channel stride and Mu [L, L]")
    n, L = E0.shape
    if out is None:
        out = torch.empty((n, L), dtype=torch.float32, device=E0.device)
    aligned = L % 4 == 0 and all(t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in (X, E0, out))
    # uniform: None = look at Mu; (alpha, beta) = the caller knows it is alpha * ones + beta * eye on the labels that matter
    # (a Mu padded with zero rows / columns for labels of probability 0: crf_module._padded_mu); False = it is not
    if uniform is None:
        uniform = _mu_uniform(Mu) if (structure and aligned and L <= 1024) else None
    elif uniform is False or not (structure and aligned and L <= 1024):
        uniform = None
    if uniform is not None:                      # Potts family: X @ Mu = alpha rowsum(X) + beta X, one streaming pass
        with torch.cuda.device(E0.device):
            _check(load_library().phl_uniform_compat_softmax(
                C.c_void_p(E0.data_ptr()), E0.stride(0), C.c_void_p(X.data_ptr()), X.stride(0), C.c_float(uniform[0]),
                C.c_float(uniform[1]), C.c_void_p(out.data_ptr()), out.stride(0), n, L, 1 if logits else 0, _stream(E0.device)))
        return out
    if aligned and L <= 256:
        mu_t = _mu_transposed(Mu, E0.device)
        if arith == "split" and hasattr(load_library(), "phl_compat_softmax_split") and load_library().phl_compat_planes_bytes(L):
            planes = _mu_planes(Mu, mu_t, E0.device)
            with torch.cuda.device(E0.device):
                _check(load_library().phl_compat_softmax_split(
                    C.c_void_p(E0.data_ptr()), E0.stride(0), C.c_void_p(X.data_ptr()), X.stride(0), C.c_void_p(mu_t.data_ptr()),
                    C.c_void_p(planes.data_ptr()), C.c_void_p(out.data_ptr()), out.stride(0), n, L, 1 if logits else 0,
                    _stream(E0.device)))
            return out
        with torch.cuda.device(E0.device):
            _check(load_library().phl_compat_softmax(
                C.c_void_p(E0.data_ptr()), E0.stride(0), C.c_void_p(X.data_ptr()), X.stride(0), C.c_void_p(mu_t.data_ptr()),
                C.c_void_p(out.data_ptr()), out.stride(0), n, L, 1 if logits else 0, _stream(E0.device)))
        return out
    G = X @ Mu.to(E0.device, torch.float32)
    if logits:
        return out.copy_(-(E0 + G))
    return softmax_neg_add(E0, G, out=out)


CRITERIA = {"AD": 0, "SD": 1, "nprod": 2}


def cost_volume(img1, img2, max_disp=None, window_size=9, criterion="AD", out=None):
    """Unary stereo cost volume on the device: the reference's ``disparity_badness(img1, img2, window_size,
    criterion)`` (crf/depth.py:36-53), returned as E_0 [h*w, max_disp] fp32 pixel-major (what
    ``mean_field_infer`` takes after the notebook's reshape, DenseCrf.ipynb cell 7).
    img1/img2: [h, w, c] float tensors or numpy arrays; max_disp defaults to w // 6 (:40)."""
    _require_gpu()
    dev = img1.device if (torch.is_tensor(img1) and img1.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(img1).to(device=dev, dtype=torch.float32).contiguous()
    b = torch.as_tensor(img2).to(device=dev, dtype=torch.float32).contiguous()
    if a.dim() == 2:
        a, b = a[..., None], b[..., None]
    if a.shape != b.shape or a.dim() != 3:
        raise ValueError(f"cost_volume: images must both be [h, w, c], got {tuple(a.shape)} and {tuple(b.shape)}")
    h, w, c = (int(v) for v in a.shape)
    L = w // 6 if max_disp is None else int(max_disp)
    crit = CRITERIA[getattr(criterion, "__name__", criterion)]
    res = torch.empty((h * w, L), dtype=torch.float32, device=dev) if out is None else out
    assert res.shape == (h * w, L) and res.stride(1) == 1 and res.dtype == torch.float32
    with torch.cuda.device(dev):
        _check(load_library().phl_cost_volume(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), h, w, c, L,
                                              int(window_size), crit, C.c_void_p(res.data_ptr()), res.stride(0) if L else 0,
                                              _stream(dev)))
    return res


def stream_copy(dst, src):
    """float4 streaming copy (HBM ceiling probe for bench.py)."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    with torch.cuda.device(src.device):
        _check(load_library().phl_stream_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel(),
                                              _stream(src.device)))
    return dst


def copy2d(dst, src):
    """dst[...] = src[...] for 2-D fp32 device tensors of equal shape and ANY strides, through the library's LDS-tiled
    transpose (phl_copy2d): the NCHW <-> pixel-major hops of the batched API."""
    assert dst.shape == src.shape and dst.dim() == 2 and dst.is_cuda and src.device == dst.device
    assert dst.dtype == torch.float32 and src.dtype == torch.float32
    with torch.cuda.device(dst.device):
        _check(load_library().phl_copy2d(C.c_void_p(src.data_ptr()), src.stride(0), src.stride(1), C.c_void_p(dst.data_ptr()),
                                         dst.stride(0), dst.stride(1), int(dst.shape[0]), int(dst.shape[1]), _stream(dst.device)))
    return dst


def expected_value(Q, labels):
    """Q @ labels (expected disparity per pixel) for CUDA fp32 Q [n, L], labels [L]."""
    if not _rowmajor(Q) or labels.dtype != torch.float32 or labels.numel() != Q.shape[1]:
        raise TypeError("expected_value: expects fp32 CUDA Q [n, L] and labels [L]")
    labels = labels.to(Q.device).contiguous()
    out = torch.empty((Q.shape[0],), dtype=torch.float32, device=Q.device)
    with torch.cuda.device(Q.device):
        _check(load_library().phl_expected_value(C.c_void_p(Q.data_ptr()), Q.stride(0), C.c_void_p(labels.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), Q.shape[0], Q.shape[1], _stream(Q.device)))
    return out


# ---------------------------------------------------------------------------------------------
# invisible lattice cache for the reference's stateless call shape
_CACHE_SIZE = int(os.environ.get("PHL_LATTICE_CACHE", "4"))
_cache = OrderedDict()
_cache_lock = threading.Lock()


# The drop-in boundary -- ``filter(src, ref)`` and everything of ``crf.*`` that reaches it through ``lattice_for`` --
# builds its lattices with the REFERENCE'S table behaviour (Lattice(reference_table=True): its duplicate vertices
# above M = 16383 included), so that what replaces ``lattice.filter`` returns what ``lattice.filter`` returned at
# any size.  PHL_REFERENCE_TABLE=0 selects the defect-free table instead (one vertex per key; a few ms less per build).
_REFERENCE_TABLE = os.environ.get("PHL_REFERENCE_TABLE", "1") not in ("", "0")


def _cache_key(ref, device=None):
    # device=None means what Lattice() resolves it to (ref's own GPU, or the current one for a CPU tensor): the same
    # lattice must be found whether a caller names that device or not (forward with device=dev, backward without)
    if device is None:
        device = ref.device if ref.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    idx = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else None)
    return (ref.device.type, ref.device.index, ref.data_ptr(), tuple(ref.shape), tuple(ref.stride()), ref._version,
            (device.type, idx))


def lattice_for(ref, device=None):
    """Cached Lattice for ``ref`` (built on ``device``; default: ref's own device, or the current one for a CPU
    tensor).  The entry keeps ``ref`` alive, so its storage address cannot be recycled while cached; an in-place
    update bumps ``ref._version`` and misses."""
    _require_gpu()
    if _CACHE_SIZE <= 0:
        return Lattice(ref, device=device, reference_table=_REFERENCE_TABLE)
    key = _cache_key(ref, device)
    with _cache_lock:
        hit = _cache.get(key)
        if hit is not None:
            _cache.move_to_end(key)
            return hit[0]
    lat = Lattice(ref, device=device, reference_table=_REFERENCE_TABLE)
    with _cache_lock:
        _cache[key] = (lat, ref)
        while len(_cache) > max(_CACHE_SIZE, _cache_floor[0]):
            _cache.popitem(last=False)
    return lat


_cache_floor = [0]      # batched_filter keeps one lattice per batch item alive across mean-field iterations


def clear_cache():
    with _cache_lock:
        _cache.clear()


def batch_devices(t=None):
    """Devices a batch of independent volumes is spread over (first row: "one image per GPU, no
    RCCL; host-side scatter of inputs / gather of outputs only").  CPU inputs -- the reference's calling
    convention, whose batch mode runs one worker process per image (crf/gaussian_matrix.py:370-377) -- go to all
    visible GPUs; tensors that already live on a GPU stay there (moving an [n, L] volume over xGMI costs more
    than filtering it) unless PHL_BATCH_DEVICES says otherwise ("all" or a comma list of indices)."""
    _require_gpu()
    env = os.environ.get("PHL_BATCH_DEVICES", "")
    ndev = torch.cuda.device_count()
    if env == "all":
        return [torch.device("cuda", i) for i in range(ndev)]
    if env:
        return [torch.device("cuda", int(i)) for i in env.split(",")]
    if t is not None and t.is_cuda:
        return [t.device]
    return [torch.device("cuda", i) for i in range(ndev)]


_batch_streams = {}


def batched_filter(srcs, refs, devices=None, subtract_input=False):
    """Independent lattice per batch item: srcs [bs, n, vd], refs [bs, n, d] (any strides) -> [bs, n, vd] on
    srcs' device.  Item i runs on devices[i % len(devices)], each device on its own side stream, so that with
    several GPUs the items (and, for CPU inputs, their PCIe copies) proceed in parallel; no collective."""
    bs = srcs.shape[0]
    assert refs.shape[0] == bs and srcs.shape[1] == refs.shape[1], "Incompatible shapes {}, and {}".format(tuple(srcs.shape), tuple(refs.shape))
    devices = [torch.device(d) for d in (devices or batch_devices(srcs))]
    _cache_floor[0] = max(_cache_floor[0], min(bs, 64))
    home = srcs.device
    out = torch.empty((bs,) + tuple(srcs.shape[1:]), dtype=torch.float32, device=home)
    used = []
    for i in range(bs):
        dev = devices[i % len(devices)]
        st = _batch_streams.get(dev)
        if st is None:
            st = _batch_streams[dev] = torch.cuda.Stream(device=dev)
        if home.type == "cuda":
            st.wait_stream(torch.cuda.current_stream(home))      # inputs may still be in flight on the caller's stream
        with torch.cuda.device(dev), torch.cuda.stream(st):
            lat = lattice_for(refs[i].detach(), device=dev)
            s = srcs[i].detach()
            res = lat.filter(s.to(dev, non_blocking=True) if s.device != dev else s, subtract_input=subtract_input)
            out[i].copy_(res, non_blocking=True)
        used.append((dev, st))
    for dev, st in used:
        if home.type == "cuda":
            torch.cuda.current_stream(home).wait_stream(st)
        else:
            st.synchronize()
    return out


def batched_filter_grad(srcs, refs, g, need_src=True, devices=None):
    """Per-item Lattice.filter_grad for a batch (the body of BatchedLatticeFilter.backward, crf/gaussian_matrix.py:402-421):
    srcs, g [bs, n, L], refs [bs, n, d] (any strides) -> (grad_srcs [bs, n, L] or None, grad_refs [bs, n, d]) on the
    inputs' devices.  Items are dealt over the same devices and side streams as batched_filter, so every item meets the
    lattice its forward pass cached.  Raises PhlError(status 7) if the fused kernels do not take the shape."""
    bs = srcs.shape[0]
    devices = [torch.device(d) for d in (devices or batch_devices(srcs))]
    home = srcs.device
    grad_refs = torch.empty(tuple(refs.shape), dtype=torch.float32, device=refs.device)
    grad_srcs = torch.empty(tuple(srcs.shape), dtype=torch.float32, device=home) if need_src else None
    used = []
    for i in range(bs):
        dev = devices[i % len(devices)]
        st = _batch_streams.get(dev)
        if st is None:
            st = _batch_streams[dev] = torch.cuda.Stream(device=dev)
        for t in (srcs, refs, g):
            if t.is_cuda:
                st.wait_stream(torch.cuda.current_stream(t.device))
        with torch.cuda.device(dev), torch.cuda.stream(st):
            r = refs[i].detach()
            lat = lattice_for(r, device=dev)
            gs, gr = lat.filter_grad(srcs[i].detach().to(dev, non_blocking=True), g[i].detach().to(dev, non_blocking=True),
                                     r.to(dev, non_blocking=True), need_src=need_src)
            grad_refs[i].copy_(gr, non_blocking=True)
            if need_src:
                grad_srcs[i].copy_(gs, non_blocking=True)
        used.append((dev, st))
    outs = [t for t in (grad_refs, grad_srcs) if t is not None]
    for dev, st in used:
        for t in outs:
            if t.is_cuda:
                torch.cuda.current_stream(t.device).wait_stream(st)
        if any(not t.is_cuda for t in outs):
            st.synchronize()
    return grad_srcs, grad_refs


def filter(src, ref):
    """Drop-in for the reference's ``lattice.filter(src, ref)``: src [n, vd], ref [n, d], fp32.

    Argument order is the reference's (src first, ref second; lattice.cpp:6)."""
    if src.shape[0] != ref.shape[0]:
        raise AssertionError("Incompatible shapes {}, and {}".format(tuple(src.shape), tuple(ref.shape)))
    return lattice_for(ref.detach()).filter(src)

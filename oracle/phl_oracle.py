"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the CPU oracle.

* ``Oracle``      : our plain-C restatement (oracle/phl_oracle.c), init-once / filter-many.
* ``reference_*`` : the reference's own engine, compiled from /root/reference by
                    oracle/build_ref.sh into oracle/_ref/libphl_ref.so (binary only; it
                    travels to the GPU box, the reference sources do not).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (depth-estimation_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(HERE, "libphl_oracle.so")
_REF_SO = os.path.join(HERE, "_ref", "libphl_ref.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i16p = C.POINTER(C.c_int16)
_f64p = C.POINTER(C.c_double)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def build_oracle(force=False):
    """Compile oracle/phl_oracle.c (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(HERE, "phl_oracle.c")
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libphl_oracle.so"], stdout=subprocess.DEVNULL)
    return _ORACLE_SO


def build_reference():
    """(Re)build oracle/_ref from /root/reference when that tree exists (container only)."""
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return _REF_SO if os.path.exists(_REF_SO) else None


_lib = None


def _oracle_lib():
    global _lib
    if _lib is None:
        build_oracle()
        lib = C.CDLL(_ORACLE_SO)
        lib.phlo_build.restype = C.c_void_p
        lib.phlo_build.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int]
        lib.phlo_free.argtypes = [C.c_void_p]
        lib.phlo_num_vertices.restype = C.c_int64
        lib.phlo_num_vertices.argtypes = [C.c_void_p]
        lib.phlo_status.argtypes = [C.c_void_p]
        lib.phlo_get_keys.argtypes = [C.c_void_p, _i16p]
        lib.phlo_get_replay.argtypes = [C.c_void_p, _i32p, _f32p]
        lib.phlo_get_neighbors.argtypes = [C.c_void_p, _i32p]
        lib.phlo_filter.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int64, C.c_int64,
                                    _f32p, C.c_int64, C.c_int64, _f32p, _f32p, _f64p]
        lib.phlo_scale_factors.argtypes = [C.c_int, _f32p]
        lib.phlo_add_vertices.argtypes = [C.c_void_p, _i16p, C.c_int64, _i32p]
        lib.phlo_splat.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int64, C.c_int64, _f32p]
        lib.phlo_blur.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.phlo_slice.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int64, C.c_int64]
        _lib = lib
    return _lib


class Oracle:
    """Lattice built once from ``ref`` [n,d] (any strides); ``filter`` many value sets."""

    def __init__(self, ref, faithful_table=False):
        """faithful_table=True reproduces the reference's stale-slot-after-grow hash defect
        (phl_oracle.c, table_lookup); only used to pin the oracle against oracle/_ref."""
        ref = np.asarray(ref)
        assert ref.dtype == np.float32 and ref.ndim == 2
        self._ref = ref  # keep alive
        self.n, self.d = ref.shape
        lib = _oracle_lib()
        rs, cs = (s // 4 for s in ref.strides)
        self._free = lib.phlo_free          # kept on the object: module globals may be gone at interpreter exit
        self._h = lib.phlo_build(_ptr(ref, _f32p), self.n, self.d, rs, cs, int(bool(faithful_table)))
        if not self._h:
            raise ValueError("phlo_build failed (d out of range?)")
        self.M = int(lib.phlo_num_vertices(self._h))
        self.status = int(lib.phlo_status(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            self._free(self._h)
            self._h = None

    def keys(self):
        out = np.empty((self.M, self.d), np.int16)
        _oracle_lib().phlo_get_keys(self._h, _ptr(out, _i16p))
        return out

    def replay(self):
        vid = np.empty((self.n, self.d + 1), np.int32)
        w = np.empty((self.n, self.d + 1), np.float32)
        _oracle_lib().phlo_get_replay(self._h, _ptr(vid, _i32p), _ptr(w, _f32p))
        return vid, w

    def neighbors(self):
        out = np.empty((self.d + 1, self.M, 2), np.int32)
        _oracle_lib().phlo_get_neighbors(self._h, _ptr(out, _i32p))
        return out

    # ---- stage API + ghost vertices (same surface as phl.Lattice; used by the row-band tests) ----
    def add_vertices(self, keys):
        keys = np.ascontiguousarray(keys, np.int16).reshape(-1, self.d)
        vid = np.empty(len(keys), np.int32)
        _oracle_lib().phlo_add_vertices(self._h, _ptr(keys, _i16p), len(keys), _ptr(vid, _i32p))
        self.M = int(_oracle_lib().phlo_num_vertices(self._h))
        return vid

    def splat(self, src):
        src = np.asarray(src, np.float32)
        vert = np.empty((self.M, src.shape[1]), np.float32)
        rs, cs = (s // 4 for s in src.strides)
        _oracle_lib().phlo_splat(self._h, _ptr(src, _f32p), src.shape[1], rs, cs, _ptr(vert, _f32p))
        return vert

    def blur(self, vert):
        vert = np.array(vert, np.float32, order="C", copy=True)
        _oracle_lib().phlo_blur(self._h, _ptr(vert, _f32p), vert.shape[1])
        return vert

    def slice(self, vert):
        vert = np.ascontiguousarray(vert, np.float32)
        out = np.empty((self.n, vert.shape[1]), np.float32)
        _oracle_lib().phlo_slice(self._h, _ptr(vert, _f32p), vert.shape[1], _ptr(out, _f32p), vert.shape[1], 1)
        return out

    def filter(self, src, stages=False, timing=False):
        src = np.asarray(src)
        assert src.dtype == np.float32 and src.ndim == 2 and src.shape[0] == self.n
        vd = src.shape[1]
        out = np.empty((self.n, vd), np.float32)
        sd = np.empty((self.M, vd), np.float32) if stages else None
        bd = np.empty((self.M, vd), np.float32) if stages else None
        t = np.zeros(3, np.float64)
        rs, cs = (s // 4 for s in src.strides)
        _oracle_lib().phlo_filter(self._h, _ptr(src, _f32p), vd, rs, cs, _ptr(out, _f32p), vd, 1,
                                  _ptr(sd, _f32p), _ptr(bd, _f32p), _ptr(t, _f64p))
        res = (out,)
        if stages:
            res += (sd, bd)
        if timing:
            res += (t,)
        return res[0] if len(res) == 1 else res


def oracle_filter(src, ref):
    """== reference ``lattice.filter(src, ref)`` (src first, ref second; lattice.cpp:6)."""
    return Oracle(ref).filter(src)


def scale_factors(d):
    sf = np.empty(d, np.float32)
    _oracle_lib().phlo_scale_factors(d, _ptr(sf, _f32p))
    return sf


# ---------------------------------------------------------------------------------------------
# the reference engine itself (binary built in the container from /root/reference)
_ref = None


def reference_available():
    return os.path.exists(_REF_SO)


def _ref_lib():
    global _ref
    if _ref is None:
        lib = C.CDLL(_REF_SO)
        lib.ref_filter.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _i16p,
                                   _f32p, _f32p, _i32p, _f32p]
        lib.ref_filter_timed.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _f64p]
        _ref = lib
    return _ref


def reference_filter(src, ref, stages=False):
    """Run the reference's own splat/blur/slice.  Returns out or a dict with stage dumps."""
    src = np.ascontiguousarray(src, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    n, d = ref.shape
    vd = src.shape[1]
    out = np.empty((n, vd), np.float32)
    M = C.c_int32(0)
    if not stages:
        _ref_lib().ref_filter(_ptr(ref, _f32p), _ptr(src, _f32p), n, d, vd, _ptr(out, _f32p),
                              C.byref(M), None, None, None, None, None)
        return out
    cap = n * (d + 1)
    keys = np.empty((cap, d), np.int16)
    sd = np.empty((cap, vd), np.float32)
    bd = np.empty((cap, vd), np.float32)
    off = np.empty((n, d + 1), np.int32)
    w = np.empty((n, d + 1), np.float32)
    _ref_lib().ref_filter(_ptr(ref, _f32p), _ptr(src, _f32p), n, d, vd, _ptr(out, _f32p), C.byref(M),
                          _ptr(keys, _i16p), _ptr(sd, _f32p), _ptr(bd, _f32p), _ptr(off, _i32p), _ptr(w, _f32p))
    m = M.value
    return dict(out=out, M=m, keys=keys[:m].copy(), splat=sd[:m].copy(), blur=bd[:m].copy(),
                replay_vid=off, replay_w=w)


def reference_filter_timed(src, ref):
    src = np.ascontiguousarray(src, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    n, d = ref.shape
    vd = src.shape[1]
    out = np.empty((n, vd), np.float32)
    M = C.c_int32(0)
    t = np.zeros(4, np.float64)
    _ref_lib().ref_filter_timed(_ptr(ref, _f32p), _ptr(src, _f32p), n, d, vd, _ptr(out, _f32p),
                                C.byref(M), _ptr(t, _f64p))
    return out, M.value, t

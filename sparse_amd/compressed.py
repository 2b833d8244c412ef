"""Host-side mirror of the reference's `compressed` module surface over the C ABI.

    c = mk_compressed("f64")            # module c = mk_compressed f64   (README.md:17)
    A = c.sr.sparse(n, m, (rs, cs, vs)) # compressed.fut:154
    y = c.sr.smvm(A, x)                 # compressed.fut:134
    B = c.sr.transpose(A)               # compressed.fut:185  (sc, zero cost)
    z = c.sc.vsmm(a, B)                 # compressed.fut:223

Names, argument order and error behaviour follow compressed.fut:24-54 and
matrix_irregular.fut:6-38; `+`/`-` are spelled add/sub.  Every call goes through
libsparse_amd.so (include/sparse_amd.h); nothing is computed in Python and there
is no fallback: a missing library raises LibraryMissing.

Arrays handed in may be numpy arrays (copied to the device), DeviceArray objects
(already there) or torch CUDA tensors (wrapped in place, zero copy).  Results are
DeviceArray objects; np.asarray(result) / result.numpy() copies to the host.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("SPARSE_AMD_LIB") or os.path.join(_HERE, "lib", "libsparse_amd.so")  # override: kernel-lab builds
_lib = None

_CT = {"f64": ctypes.c_double, "f32": ctypes.c_float, "i32": ctypes.c_int32, "i64": ctypes.c_int64}
_NP = {"f64": np.float64, "f32": np.float32, "i32": np.int32, "i64": np.int64}
_VP = ctypes.c_void_p
_I64 = ctypes.c_int64


class LibraryMissing(RuntimeError):
    pass


class SparseError(RuntimeError):
    """A non-zero return from an entry point (the reference's failed `assert`)."""


def library_path():
    return _LIBPATH


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own libamdhip64.so (soname
    libamdhip64.so.7) and asks for it by file name; libsparse_amd.so asks for the soname.  If torch comes
    first the loader hands its copy to us and all is well; if we came first, torch would bring in a second
    runtime (and a second HSA runtime, which then finds no GPUs, and whose streams mean nothing to ours).
    So when a torch wheel is installed and not yet loaded, map its runtime first; ours then binds to it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise LibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _LIBPATH)
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(_LIBPATH)
        lib.futhark_context_config_new.restype = _VP
        lib.futhark_context_new.restype = _VP
        lib.futhark_context_new.argtypes = [_VP]
        lib.futhark_context_get_error.restype = _VP
        lib.futhark_context_report.restype = _VP
        lib.sparse_amd_context_get_stream.restype = _VP
        lib.sparse_amd_context_get_option.restype = _I64
        lib.sparse_amd_version.restype = ctypes.c_char_p
        for name in ("i64_1d", "i32_1d", "f32_1d", "f64_1d", "i32_2d", "f32_2d", "f64_2d"):
            for pre in ("futhark_new_", "futhark_new_raw_", "sparse_amd_wrap_", "futhark_values_raw_",
                        "futhark_shape_"):
                getattr(lib, pre + name).restype = _VP
        _lib = lib
    return _lib


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [_VP]


class Context:
    """futhark_context over one HIP device."""

    def __init__(self, device=0, logging=False):
        L = load_library()
        self.L = L
        cfg = L.futhark_context_config_new()
        L.futhark_context_config_set_device(_VP(cfg), ("#%d" % device).encode())
        L.futhark_context_config_set_logging(_VP(cfg), int(logging))
        self.h = _VP(L.futhark_context_new(_VP(cfg)))
        L.futhark_context_config_free(_VP(cfg))
        self.device = device
        err = self.error()
        if err:
            raise SparseError(err)

    def error(self):
        p = self.L.futhark_context_get_error(self.h)
        if not p:
            return None
        s = ctypes.string_at(p).decode()
        _libc.free(_VP(p))
        return s

    def check(self, rc):
        if rc != 0:
            raise SparseError(self.error() or "entry point failed (rc=%d)" % rc)

    def sync(self):
        self.check(self.L.futhark_context_sync(self.h))

    def clear_caches(self):
        self.check(self.L.futhark_context_clear_caches(self.h))

    def report(self):
        p = self.L.futhark_context_report(self.h)
        s = ctypes.string_at(p).decode()
        _libc.free(_VP(p))
        return s

    def set_stream(self, hip_stream):
        self.check(self.L.sparse_amd_context_set_stream(self.h, _VP(hip_stream)))

    def get_stream(self):
        return self.L.sparse_amd_context_get_stream(self.h)

    def set_option(self, key, value):
        self.check(self.L.sparse_amd_context_set_option(self.h, key.encode(), _I64(int(value))))

    def get_option(self, key):
        return int(self.L.sparse_amd_context_get_option(self.h, key.encode()))

    def measure_bandwidth(self, read_bytes_per_group, write_bytes_per_group, groups):
        """(GB/s, microseconds per launch) of a streaming read/write mix with no arithmetic
        (sparse_amd_context_measure_bandwidth): the measured ceiling a kernel of that access mix can be held against."""
        gbs, us = ctypes.c_double(), ctypes.c_double()
        self.check(self.L.sparse_amd_context_measure_bandwidth(self.h, _I64(read_bytes_per_group), _I64(write_bytes_per_group),
                                                               _I64(groups), ctypes.byref(gbs), ctypes.byref(us)))
        return gbs.value, us.value

    def close(self):
        if self.h:
            self.L.futhark_context_free(self.h)
            self.h = None

    # ---- arrays
    def array(self, a, dtype, rank=1):
        """numpy / DeviceArray / torch CUDA tensor -> DeviceArray of `dtype`."""
        if isinstance(a, DeviceArray):
            assert a.dtype == dtype and a.rank == rank, (a.dtype, dtype, a.rank, rank)
            return a
        if hasattr(a, "data_ptr") and getattr(a, "is_cuda", False):   # torch tensor on the GPU
            assert a.is_contiguous() and a.dim() == rank
            assert a.element_size() == np.dtype(_NP[dtype]).itemsize, "tensor dtype does not match %s" % dtype
            name = "%s_%dd" % (dtype, rank)
            h = getattr(self.L, "sparse_amd_wrap_" + name)(self.h, _VP(a.data_ptr()), *[_I64(int(s)) for s in a.shape])
            d = DeviceArray(self, h, dtype, rank)
            d._keepalive = a
            return d
        a = np.ascontiguousarray(np.asarray(a, dtype=_NP[dtype]))
        if rank == 1:
            a = a.reshape(-1)
        assert a.ndim == rank
        name = "%s_%dd" % (dtype, rank)
        h = getattr(self.L, "futhark_new_" + name)(self.h, a.ctypes.data_as(_VP), *[_I64(s) for s in a.shape])
        if not h:
            raise SparseError(self.error() or "futhark_new_%s failed" % name)
        return DeviceArray(self, h, dtype, rank)

    def _out(self, h, dtype, rank=1):
        return DeviceArray(self, h.value if isinstance(h, _VP) else h, dtype, rank)


class DeviceArray:
    """A futhark_<t>_<r>d owned by a Context."""

    def __init__(self, ctx, handle, dtype, rank):
        self.ctx, self.h, self.dtype, self.rank = ctx, _VP(handle), dtype, rank
        self.name = "%s_%dd" % (dtype, rank)

    @property
    def shape(self):
        p = getattr(self.ctx.L, "futhark_shape_" + self.name)(self.ctx.h, self.h)
        return tuple(ctypes.cast(p, ctypes.POINTER(_I64))[i] for i in range(self.rank))

    @property
    def ptr(self):
        """Device pointer (futhark_values_raw_*)."""
        return getattr(self.ctx.L, "futhark_values_raw_" + self.name)(self.ctx.h, self.h) or 0

    def numpy(self):
        out = np.empty(self.shape, _NP[self.dtype])
        self.ctx.check(getattr(self.ctx.L, "futhark_values_" + self.name)(self.ctx.h, self.h, out.ctypes.data_as(_VP)))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.shape[0]

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": np.dtype(_NP[self.dtype]).str, "data": (self.ptr, False),
                "version": 2, "strides": None}

    def free(self):
        if self.h and self.ctx.h:
            getattr(self.ctx.L, "futhark_free_" + self.name)(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Matrix:
    """futhark_opaque_csr_<t> / futhark_opaque_csc_<t>."""

    def __init__(self, ctx, handle, kind, dtype):
        self.ctx, self.h, self.kind, self.dtype = ctx, _VP(handle), kind, dtype

    def stored_dims(self):
        n, m, s = _I64(), _I64(), _I64()
        f = getattr(self.ctx.L, "sparse_amd_%s_%s_dims" % (self.kind, self.dtype))
        self.ctx.check(f(self.ctx.h, self.h, ctypes.byref(n), ctypes.byref(m), ctypes.byref(s)))
        return n.value, m.value, s.value

    def fields(self):
        """(row_off[n] inclusive, col_idx[nnz] i64, vals[nnz]) of the stored record (compressed.fut:93-96)."""
        ro, ci, va = _VP(), _VP(), _VP()
        f = getattr(self.ctx.L, "sparse_amd_%s_%s_fields" % (self.kind, self.dtype))
        self.ctx.check(f(self.ctx.h, ctypes.byref(ro), ctypes.byref(ci), ctypes.byref(va), self.h))
        return (self.ctx._out(ro, "i64"), self.ctx._out(ci, "i64"), self.ctx._out(va, self.dtype))

    def store(self):
        """The handle's bytes (futhark_store_opaque_*): n, m, row_off, col_idx, vals of the stored record."""
        p, n = _VP(), ctypes.c_size_t()
        f = getattr(self.ctx.L, "futhark_store_opaque_%s_%s" % (self.kind, self.dtype))
        self.ctx.check(f(self.ctx.h, self.h, ctypes.byref(p), ctypes.byref(n)))
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            _libc.free(p)

    def free(self):
        if self.h and self.ctx.h:
            getattr(self.ctx.L, "futhark_free_opaque_%s_%s" % (self.kind, self.dtype))(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Module:
    """Shared body of `sr` and `sc` (matrix_irregular.fut:6-38)."""
    kind = None

    def __init__(self, parent):
        self.p = parent
        self.ctx = parent.ctx
        self.t = parent.t

    def _f(self, op):
        return getattr(self.ctx.L, "futhark_entry_%s_%s_%s" % (self.kind, self.t, op))

    def _new(self, h, kind=None):
        return Matrix(self.ctx, h.value, kind or self.kind, self.t)

    def _chk(self, A, kind=None):
        assert isinstance(A, Matrix) and A.kind == (kind or self.kind) and A.dtype == self.t, \
            "expected a %s %s matrix" % (kind or self.kind, self.t)
        return A.h

    def zero(self, n, m):
        out = _VP()
        self.ctx.check(self._f("zero")(self.ctx.h, ctypes.byref(out), _I64(n), _I64(m)))
        return self._new(out)

    def eye(self, n, m):
        out = _VP()
        self.ctx.check(self._f("eye")(self.ctx.h, ctypes.byref(out), _I64(n), _I64(m)))
        return self._new(out)

    def diag(self, v):
        v = self.ctx.array(v, self.t)
        out = _VP()
        self.ctx.check(self._f("diag")(self.ctx.h, ctypes.byref(out), v.h))
        return self._new(out)

    def dense(self, A):
        out = _VP()
        self.ctx.check(self._f("dense")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self.ctx._out(out, self.t, 2)

    def scale(self, v, A):
        out = _VP()
        self.ctx.check(self._f("scale")(self.ctx.h, ctypes.byref(out), _CT[self.t](v), self._chk(A)))
        return self._new(out)

    def sparse(self, n, m, coo):
        rs, cs, vs = coo
        rs = self.ctx.array(rs, "i64")
        cs = self.ctx.array(cs, "i64")
        vs = self.ctx.array(vs, self.t)
        out = _VP()
        self.ctx.check(self._f("sparse")(self.ctx.h, ctypes.byref(out), _I64(n), _I64(m), rs.h, cs.h, vs.h))
        return self._new(out)

    def prepare(self, A, expected_products):
        """How many products the caller will run with A (sparse_amd_<kind>_<t>_prepare): 16 or more builds the
        record's faster SpMV form now, fewer pins the form that needs nothing built.  No counterpart in the reference."""
        f = getattr(self.ctx.L, "sparse_amd_%s_%s_prepare" % (self.kind, self.t))
        self.ctx.check(f(self.ctx.h, self._chk(A), _I64(int(expected_products))))
        return A

    def nnz(self, A):
        out = _I64()
        self.ctx.check(self._f("nnz")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return out.value

    def coo(self, A):
        r, c, v = _VP(), _VP(), _VP()
        self.ctx.check(self._f("coo")(self.ctx.h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(v), self._chk(A)))
        return (self.ctx._out(r, "i64"), self.ctx._out(c, "i64"), self.ctx._out(v, self.t))

    def restore(self, data):
        """A matrix from Matrix.store() bytes (futhark_restore_opaque_*); the record is validated."""
        f = getattr(self.ctx.L, "futhark_restore_opaque_%s_%s" % (self.kind, self.t))
        f.restype = _VP
        h = f(self.ctx.h, ctypes.c_char_p(bytes(data)))
        if not h:
            raise SparseError(self.ctx.error() or "restore failed")
        return Matrix(self.ctx, h, self.kind, self.t)

    def add(self, A, B):
        out = _VP()
        self.ctx.check(self._f("add")(self.ctx.h, ctypes.byref(out), self._chk(A), self._chk(B)))
        return self._new(out)

    def sub(self, A, B):
        out = _VP()
        self.ctx.check(self._f("sub")(self.ctx.h, ctypes.byref(out), self._chk(A), self._chk(B)))
        return self._new(out)


class _SR(_Module):
    kind = "csr"

    def transpose(self, A):
        """sr[n][m] -> sc[m][n]; moves no data (compressed.fut:185-186)."""
        out = _VP()
        self.ctx.check(self._f("transpose")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "csc")

    def smvm(self, A, v):
        v = self.ctx.array(v, self.t)
        out = _VP()
        self.ctx.check(self._f("smvm")(self.ctx.h, ctypes.byref(out), self._chk(A), v.h))
        return self.ctx._out(out, self.t)

    def smvm_into(self, y, A, v, rows=None):
        """sr.smvm into caller-owned storage (sparse_amd_csr_*_smvm_into): y = A v, or with `rows`
        (distinct indices into y, one per row of A) y[rows[i]] += (A v)[i].  Pass DeviceArray
        objects made once with Context.array to keep the per-call host cost to one C call."""
        y = self.ctx.array(y, self.t)
        v = self.ctx.array(v, self.t)
        rows = self.ctx.array(rows, "i64") if rows is not None else None
        f = getattr(self.ctx.L, "sparse_amd_csr_%s_smvm_into" % self.t)
        self.ctx.check(f(self.ctx.h, y.h, self._chk(A), v.h, rows.h if rows is not None else None))
        return y

    def smvm_into_noalloc(self, y, A, v):
        """y = A v through the entry point that allocates nothing (sparse_amd_csr_*_smvm_into_noalloc);
        y and v must already be DeviceArray objects or torch CUDA tensors."""
        y = self.ctx.array(y, self.t)
        v = self.ctx.array(v, self.t)
        f = getattr(self.ctx.L, "sparse_amd_csr_%s_smvm_into_noalloc" % self.t)
        self.ctx.check(f(self.ctx.h, y.h, self._chk(A), v.h))
        return y

    def smm(self, A, B):
        """Sparse x dense panel, C[:,j] = smvm A B[:,j] (build-defined; see include/sparse_amd.h)."""
        B = self.ctx.array(B, self.t, 2)
        out = _VP()
        self.ctx.check(self._f("smm")(self.ctx.h, ctypes.byref(out), self._chk(A), B.h))
        return self.ctx._out(out, self.t, 2)

    def to_sc(self, A):
        """Physical conversion sc.sparse n m (sr.coo A)."""
        out = _VP()
        self.ctx.check(self._f("to_csc")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "csc")

    def from_fields(self, n, m, row_off, col_idx, vals):
        ro = self.ctx.array(row_off, "i64")
        ci = self.ctx.array(col_idx, "i64")
        va = self.ctx.array(vals, self.t)
        out = _VP()
        f = getattr(self.ctx.L, "sparse_amd_csr_%s_from_fields" % self.t)
        self.ctx.check(f(self.ctx.h, ctypes.byref(out), _I64(n), _I64(m), ro.h, ci.h, va.h))
        return self._new(out)


class _SC(_Module):
    kind = "csc"

    def transpose(self, A):
        """sc[n][m] -> sr[m][n]; moves no data (compressed.fut:220-221)."""
        out = _VP()
        self.ctx.check(self._f("transpose")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "csr")

    def vsmm(self, a, B):
        a = self.ctx.array(a, self.t)
        out = _VP()
        self.ctx.check(self._f("vsmm")(self.ctx.h, ctypes.byref(out), a.h, self._chk(B)))
        return self.ctx._out(out, self.t)

    def to_sr(self, A):
        """Physical conversion sr.sparse n m (sc.coo A)."""
        out = _VP()
        self.ctx.check(self._f("to_csr")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "csr")


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class mk_compressed:
    """`module c = mk_compressed T` (compressed.fut:61) for T in f64, f32, i32."""

    def __init__(self, T, ctx=None):
        assert T in ("f64", "f32", "i32"), T
        self.t = T
        self.ctx = ctx if ctx is not None else default_context()
        self.sr = _SR(self)
        self.sc = _SC(self)

    def smsmm(self, A, B):
        """sr[n][m] -> sc[m][k] -> sr[n][k] (compressed.fut:268-331)."""
        out = _VP()
        f = getattr(self.ctx.L, "futhark_entry_smsmm_%s" % self.t)
        self.ctx.check(f(self.ctx.h, ctypes.byref(out), self.sr._chk(A), self.sc._chk(B)))
        return Matrix(self.ctx, out.value, "csr", self.t)

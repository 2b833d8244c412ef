"""ctypes mirror of the native multi-GPU API (include/sparse_amd.h section 6; sparse_amd/csrc/group.hpp): one
process that owns several devices.  Thin on purpose: partitioning, assembly, the exchange of x and the products
all happen inside libsparse_amd.so.

    g = Group([0, 1, 2, 3])                       # transport: RCCL when the devices are distinct
    A = g.sparse(n, m, (rs, cs, vs))              # host triples -> row blocks on the members
    x = g.vector(A, x_host)                       # cut like A's columns
    y = g.vector(A, None, like_rows=True)
    g.smvm(y, A, x); y_host = y.numpy()
"""
import ctypes

import numpy as np

from .compressed import SparseError, _I64, _VP, _libc, load_library


class Group:
    def __init__(self, devices, transport=None):
        self.L = L = load_library()
        L.sparse_amd_group_get_error.restype = _VP
        L.sparse_amd_group_transport.restype = ctypes.c_char_p
        L.sparse_amd_group_context.restype = _VP
        L.sparse_amd_dcsr_f64_member.restype = _VP
        L.sparse_amd_dvec_f64_member_ptr.restype = _VP
        devs = (ctypes.c_int * len(devices))(*devices)
        h = _VP()
        rc = L.sparse_amd_group_new(ctypes.byref(h), len(devices), devs, transport.encode() if transport else None)
        self.h = h
        self.size = len(devices)
        if rc != 0:
            msg = self.error() if h else "group creation failed"
            if h:
                L.sparse_amd_group_free(h)
                self.h = None
            raise SparseError(msg)

    def error(self):
        p = self.L.sparse_amd_group_get_error(self.h)
        if not p:
            return None
        s = ctypes.string_at(p).decode()
        _libc.free(_VP(p))
        return s

    def check(self, rc):
        if rc != 0:
            raise SparseError(self.error() or "group call failed (rc=%d)" % rc)

    @property
    def transport(self):
        return self.L.sparse_amd_group_transport(self.h).decode()

    def sync(self):
        self.check(self.L.sparse_amd_group_sync(self.h))

    def sparse(self, n, m, coo):
        rs, cs, vs = (np.ascontiguousarray(a, dtype=t) for a, t in zip(coo, (np.int64, np.int64, np.float64)))
        assert len(rs) == len(cs) == len(vs)
        out = _VP()
        self.check(self.L.sparse_amd_dcsr_f64_sparse(self.h, ctypes.byref(out), _I64(n), _I64(m), rs.ctypes.data_as(_VP),
                                                     cs.ctypes.data_as(_VP), vs.ctypes.data_as(_VP), _I64(len(rs))))
        return DistMatrix(self, out, n, m)

    def vector(self, A, host=None, like_rows=False):
        out = _VP()
        if host is not None:
            host = np.ascontiguousarray(host, dtype=np.float64)
            assert len(host) == (A.n if like_rows else A.m)
        self.check(self.L.sparse_amd_dvec_f64_new(self.h, ctypes.byref(out), A.h, int(like_rows),
                                                  host.ctypes.data_as(_VP) if host is not None else None))
        return DistVector(self, out, A.n if like_rows else A.m)

    def smvm(self, y, A, x):
        self.check(self.L.sparse_amd_dcsr_f64_smvm(self.h, y.h, A.h, x.h))
        return y

    def close(self):
        if self.h:
            self.L.sparse_amd_group_free(self.h)
            self.h = None


class DistMatrix:
    def __init__(self, g, h, n, m):
        self.g, self.h, self.n, self.m = g, h, n, m

    def bounds(self):
        rb = (ctypes.c_int64 * (self.g.size + 1))()
        cb = (ctypes.c_int64 * (self.g.size + 1))()
        self.g.check(self.g.L.sparse_amd_dcsr_f64_bounds(self.g.h, self.h, rb, cb))
        return list(rb), list(cb)

    def member_fields(self, k):
        """(row_off, col_idx, vals) of member k's row block, through the member's own context."""
        L = self.g.L
        ctx = _VP(L.sparse_amd_group_context(self.g.h, k))
        a = _VP(L.sparse_amd_dcsr_f64_member(self.g.h, self.h, k))
        ro, ci, va = _VP(), _VP(), _VP()
        if L.sparse_amd_csr_f64_fields(ctx, ctypes.byref(ro), ctypes.byref(ci), ctypes.byref(va), a) != 0:
            raise SparseError("fields failed on member %d" % k)
        out = []
        L.futhark_shape_i64_1d.restype = _VP
        L.futhark_shape_f64_1d.restype = _VP
        for h, name, dt in ((ro, "i64", np.int64), (ci, "i64", np.int64), (va, "f64", np.float64)):
            shp = ctypes.cast(getattr(L, "futhark_shape_%s_1d" % name)(ctx, h), ctypes.POINTER(ctypes.c_int64))[0]
            arr = np.empty(shp, dt)
            if getattr(L, "futhark_values_%s_1d" % name)(ctx, h, arr.ctypes.data_as(_VP)) != 0:
                raise SparseError("values failed on member %d" % k)
            getattr(L, "futhark_free_%s_1d" % name)(ctx, h)
            out.append(arr)
        return tuple(out)

    def free(self):
        if self.h and self.g.h:
            self.g.L.sparse_amd_dcsr_f64_free(self.g.h, self.h)
        self.h = None


class DistVector:
    def __init__(self, g, h, length):
        self.g, self.h, self.length = g, h, length

    def numpy(self):
        out = np.empty(self.length, np.float64)
        self.g.check(self.g.L.sparse_amd_dvec_f64_values(self.g.h, self.h, out.ctypes.data_as(_VP)))
        return out

    def free(self):
        if self.h and self.g.h:
            self.g.L.sparse_amd_dvec_f64_free(self.g.h, self.h)
        self.h = None

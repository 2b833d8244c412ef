"""Host-side mirror of the reference's module type `mono` (mono.fut:24-57):

    mo = mk_mono("f64")                 # module mo = mk_mono f64   (mono.fut:61)
    A = mo.sr.sparse(n, m, (rs, cs, vs))    # mono.fut:119: at most one element per row
    y = mo.sr.smvm(A, x)                # mono.fut:154
    z = mo.sr.vsmm(v, A)                # mono.fut:157
    C = mo.sr.dmsmm(D, A)               # mono.fut:161

Same conventions as sparse_amd.compressed (every call goes through libsparse_amd.so;
no fallback).  `sparse` raises SparseError for an index out of range and for two
elements in one row (the reference's two asserts, mono.fut:120-126); `+`/`-` (add/sub)
raise when the operands' column indices differ (mono.fut:141, :147).
"""
import ctypes

from .compressed import _VP, Matrix, _Module, default_context


class MonoMatrix(Matrix):
    """futhark_opaque_msr_<t> / futhark_opaque_msc_<t>."""

    def stored_dims(self):
        raise NotImplementedError("mono records have no stored-count; use fields()")

    def fields(self):
        """(col_idx[n], vals[n]) of the stored record (mono.fut:95)."""
        ci, va = _VP(), _VP()
        f = getattr(self.ctx.L, "sparse_amd_%s_%s_fields" % (self.kind, self.dtype))
        self.ctx.check(f(self.ctx.h, ctypes.byref(ci), ctypes.byref(va), self.h))
        return self.ctx._out(ci, "i64"), self.ctx._out(va, self.dtype)


class _MonoModule(_Module):
    def _new(self, h, kind=None):
        return MonoMatrix(self.ctx, h.value, kind or self.kind, self.t)


class _MSR(_MonoModule):
    kind = "msr"

    def transpose(self, A):
        """sr[n][m] -> sc[m][n]; moves no data (mono.fut:151-152)."""
        out = _VP()
        self.ctx.check(self._f("transpose")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "msc")

    def smvm(self, A, v):
        v = self.ctx.array(v, self.t)
        out = _VP()
        self.ctx.check(self._f("smvm")(self.ctx.h, ctypes.byref(out), self._chk(A), v.h))
        return self.ctx._out(out, self.t)

    def vsmm(self, v, A):
        v = self.ctx.array(v, self.t)
        out = _VP()
        self.ctx.check(self._f("vsmm")(self.ctx.h, ctypes.byref(out), v.h, self._chk(A)))
        return self.ctx._out(out, self.t)

    def dmsmm(self, D, S):
        D = self.ctx.array(D, self.t, 2)
        out = _VP()
        self.ctx.check(self._f("dmsmm")(self.ctx.h, ctypes.byref(out), D.h, self._chk(S)))
        return self.ctx._out(out, self.t, 2)


class _MSC(_MonoModule):
    kind = "msc"

    def transpose(self, A):
        """sc[n][m] -> sr[m][n]; moves no data (mono.fut:197-198)."""
        out = _VP()
        self.ctx.check(self._f("transpose")(self.ctx.h, ctypes.byref(out), self._chk(A)))
        return self._new(out, "msr")

    def vsmm(self, a, B):
        a = self.ctx.array(a, self.t)
        out = _VP()
        self.ctx.check(self._f("vsmm")(self.ctx.h, ctypes.byref(out), a.h, self._chk(B)))
        return self.ctx._out(out, self.t)


class mk_mono:
    """`module mo = mk_mono T` (mono.fut:61) for T in f64, f32, i32."""

    def __init__(self, T, ctx=None):
        assert T in ("f64", "f32", "i32"), T
        self.t = T
        self.ctx = ctx if ctx is not None else default_context()
        self.sr = _MSR(self)
        self.sc = _MSC(self)

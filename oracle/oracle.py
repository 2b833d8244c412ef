"""ctypes binding over oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

The API mirrors the reference's module layout so tests read like
compressed_test.fut:   O = Oracle("i32");  A = O.sr_sparse(n, m, rs, cs, vs)
A matrix is a plain dict {n, m, nnz, row_off, col_idx, vals} holding the
reference's record fields (compressed.fut:93-96): `row_off` is the INCLUSIVE
prefix sum of length n.  For `sc` matrices the dict holds the *stored*
sr.mat[m][n] (compressed.fut:226), with n/m the stored dimensions.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_CT = {"f64": ctypes.c_double, "f32": ctypes.c_float, "i32": ctypes.c_int32}
_NP = {"f64": np.float64, "f32": np.float32, "i32": np.int32}


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("compressed_oracle.c", "compressed_oracle.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_splitmix64.restype = ctypes.c_uint64
        _LIB.oracle_splitmix64.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def splitmix64(seed, index):
    return int(lib().oracle_splitmix64(seed, index))


class Oracle:
    def __init__(self, dtype):
        assert dtype in _CT, dtype
        self.dtype = dtype
        self.np = _NP[dtype]
        self.ct = _CT[dtype]
        self.L = lib()

    def _f(self, name, restype=ctypes.c_int):
        f = getattr(self.L, "%s_%s" % (name, self.dtype))
        f.restype = restype
        return f

    def _vals(self, a):
        return np.ascontiguousarray(np.asarray(a, dtype=self.np))

    @staticmethod
    def _mat(n, m, nnz, ro, ci, va):
        return {"n": int(n), "m": int(m), "nnz": int(nnz), "row_off": ro,
                "col_idx": ci[:nnz].copy(), "vals": va[:nnz].copy()}

    # -- construction -------------------------------------------------------
    def sr_sparse(self, n, m, rs, cs, vs):
        rs, cs, vs = _i64(rs), _i64(cs), self._vals(vs)
        k = len(rs)
        ro = np.zeros(max(n, 0), np.int64)
        ci = np.zeros(max(k, 1), np.int64)
        va = np.zeros(max(k, 1), self.np)
        nnz = ctypes.c_int64(0)
        rc = self._f("oracle_sr_sparse")(ctypes.c_int64(n), ctypes.c_int64(m), ctypes.c_int64(k),
                                         _p(rs), _p(cs), _p(vs), _p(ro), _p(ci), _p(va),
                                         ctypes.byref(nnz))
        if rc != 0:
            raise ValueError("oracle: index out of bounds (compressed.fut:156 assert)")
        return self._mat(n, m, nnz.value, ro, ci, va)

    def sc_sparse(self, n, m, rs, cs, vs):
        """sc.sparse n m coo = sr.sparse m n (swap coo) (compressed.fut:211-212)."""
        return self.sr_sparse(m, n, cs, rs, vs)

    def sr_zero(self, n, m):
        return self._mat(n, m, 0, np.zeros(n, np.int64), np.zeros(1, np.int64), np.zeros(1, self.np))

    def sr_eye(self, n, m):
        e = min(n, m)
        ro = np.zeros(n, np.int64)
        ci = np.zeros(max(e, 1), np.int64)
        va = np.zeros(max(e, 1), self.np)
        self._f("oracle_sr_eye", None)(ctypes.c_int64(n), ctypes.c_int64(m), _p(ro), _p(ci), _p(va))
        return self._mat(n, m, e, ro, ci, va)

    def sc_eye(self, n, m):
        return self.sr_eye(m, n)

    def sr_diag(self, v):
        v = self._vals(v)
        n = len(v)
        ro = np.zeros(n, np.int64)
        ci = np.zeros(max(n, 1), np.int64)
        va = np.zeros(max(n, 1), self.np)
        self._f("oracle_sr_diag", None)(ctypes.c_int64(n), _p(v), _p(ro), _p(ci), _p(va))
        return self._mat(n, n, n, ro, ci, va)

    # -- products -----------------------------------------------------------
    def sr_smvm(self, A, v, threads=0):
        v = self._vals(v)
        assert len(v) == A["m"]
        out = np.zeros(A["n"], self.np)
        if threads:
            self._f("oracle_sr_smvm_mt", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]),
                                              _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]),
                                              _p(v), _p(out), ctypes.c_int(threads))
        else:
            self._f("oracle_sr_smvm", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]),
                                           _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]),
                                           _p(v), _p(out))
        return out

    def sc_vsmm(self, a, B):
        """vsmm a B = sr.smvm (transpose B) a (compressed.fut:223-224); B is the stored sr.mat."""
        return self.sr_smvm(B, a)

    def sr_smm(self, A, B):
        B = np.ascontiguousarray(np.asarray(B, dtype=self.np))
        assert B.shape[0] == A["m"]
        k = B.shape[1]
        C = np.zeros((A["n"], k), self.np)
        self._f("oracle_sr_smm", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]), ctypes.c_int64(k),
                                      _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]), _p(B), _p(C))
        return C

    # -- observation --------------------------------------------------------
    def sr_coo(self, A):
        nnz = A["nnz"]
        r = np.zeros(max(nnz, 1), np.int64)
        c = np.zeros(max(nnz, 1), np.int64)
        v = np.zeros(max(nnz, 1), self.np)
        self._f("oracle_sr_coo", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(nnz), _p(A["row_off"]),
                                      _p(A["col_idx"]), _p(A["vals"]), _p(r), _p(c), _p(v))
        return r[:nnz], c[:nnz], v[:nnz]

    def sc_coo(self, A):
        r, c, v = self.sr_coo(A)
        return c, r, v

    def sr_dense(self, A):
        out = np.zeros((A["n"], A["m"]), self.np)
        if A["n"] * A["m"] > 0:
            self._f("oracle_sr_dense", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]),
                                            _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]), _p(out))
        return out

    def sc_dense(self, A):
        out = np.zeros((A["m"], A["n"]), self.np)
        if A["n"] * A["m"] > 0:
            self._f("oracle_sc_dense", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]),
                                            _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]), _p(out))
        return out

    def sr_nnz(self, A):
        return int(self._f("oracle_sr_nnz", ctypes.c_int64)(ctypes.c_int64(A["nnz"]), _p(A["vals"])))

    def sr_scale(self, v, A):
        out = np.zeros(max(A["nnz"], 1), self.np)
        self._f("oracle_sr_scale", None)(ctypes.c_int64(A["nnz"]), self.ct(v), _p(A["vals"]), _p(out))
        return self._mat(A["n"], A["m"], A["nnz"], A["row_off"].copy(), A["col_idx"], out)

    # -- algebra ------------------------------------------------------------
    def _addsub(self, sub, A, B):
        assert A["n"] == B["n"] and A["m"] == B["m"]
        tot = A["nnz"] + B["nnz"]
        ro = np.zeros(A["n"], np.int64)
        ci = np.zeros(max(tot, 1), np.int64)
        va = np.zeros(max(tot, 1), self.np)
        nnz = ctypes.c_int64(0)
        rc = self._f("oracle_sr_addsub")(ctypes.c_int(sub), ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]),
                                         ctypes.c_int64(A["nnz"]), _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]),
                                         ctypes.c_int64(B["nnz"]), _p(B["row_off"]), _p(B["col_idx"]), _p(B["vals"]),
                                         _p(ro), _p(ci), _p(va), ctypes.byref(nnz))
        assert rc == 0
        return self._mat(A["n"], A["m"], nnz.value, ro, ci, va)

    def sr_add(self, A, B):
        return self._addsub(0, A, B)

    def sr_sub(self, A, B):
        return self._addsub(1, A, B)

    def sr_to_sc(self, A):
        """Physical conversion sc.sparse n m (sr.coo A): returns the stored sr.mat[m][n]."""
        ro = np.zeros(A["m"], np.int64)
        ci = np.zeros(max(A["nnz"], 1), np.int64)
        va = np.zeros(max(A["nnz"], 1), self.np)
        nnz = ctypes.c_int64(0)
        rc = self._f("oracle_sr_to_sc")(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]), ctypes.c_int64(A["nnz"]),
                                        _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]),
                                        _p(ro), _p(ci), _p(va), ctypes.byref(nnz))
        assert rc == 0
        return self._mat(A["m"], A["n"], nnz.value, ro, ci, va)

    def smsmm(self, A, B):
        """A: sr[n][m]; B: sc[m][k] given as its stored sr.mat[k][m] (compressed.fut:268-331)."""
        n, m, k = A["n"], A["m"], B["n"]
        assert B["m"] == m
        cap = max(A["nnz"] * k + B["nnz"] * n, 1)
        ro = np.zeros(n, np.int64)
        ci = np.zeros(cap, np.int64)
        va = np.zeros(cap, self.np)
        nnz = ctypes.c_int64(0)
        rc = self._f("oracle_smsmm")(ctypes.c_int64(n), ctypes.c_int64(m), ctypes.c_int64(k),
                                     ctypes.c_int64(A["nnz"]), _p(A["row_off"]), _p(A["col_idx"]), _p(A["vals"]),
                                     ctypes.c_int64(B["nnz"]), _p(B["row_off"]), _p(B["col_idx"]), _p(B["vals"]),
                                     _p(ro), _p(ci), _p(va), ctypes.c_int64(cap), ctypes.byref(nnz))
        assert rc == 0, rc
        return self._mat(n, k, nnz.value, ro, ci, va)

    # ===================================================================== mono
    # Records are dicts {"n","m","col_idx"[n],"vals"[n]} of the STORED sr.mat (mono.fut:95);
    # an sc matrix of shape n x m is the stored record with n and m swapped (mono.fut:203).
    @staticmethod
    def _mono(n, m, ci, va):
        return {"n": int(n), "m": int(m), "col_idx": ci, "vals": va}

    def msr_sparse(self, n, m, rs, cs, vs):
        rs, cs, vs = _i64(rs), _i64(cs), self._vals(vs)
        ci = np.zeros(max(n, 0), np.int64)
        va = np.zeros(max(n, 0), self.np)
        rc = self._f("oracle_msr_sparse")(ctypes.c_int64(n), ctypes.c_int64(m), ctypes.c_int64(len(rs)),
                                          _p(rs), _p(cs), _p(vs), _p(ci), _p(va))
        if rc == 1:
            raise ValueError("oracle: index out of bounds (mono.fut:120-121 assert)")
        if rc == 2:
            raise ValueError("oracle: two elements in one row (mono.fut:122-126 assert)")
        return self._mono(n, m, ci, va)

    def msc_sparse(self, n, m, rs, cs, vs):
        """mono.fut:188-189: map swap coo |> sr.sparse m n."""
        return self.msr_sparse(m, n, cs, rs, vs)

    def msr_zero(self, n, m):       # mono.fut:97-100
        return self._mono(n, m, np.zeros(n, np.int64), np.zeros(n, self.np))

    def msr_eye(self, n, m):        # mono.fut:102-105: col_idx = iota n whatever m is
        return self._mono(n, m, np.arange(n, dtype=np.int64), np.ones(n, self.np))

    def msr_diag(self, v):          # mono.fut:107-110
        v = self._vals(v)
        return self._mono(len(v), len(v), np.arange(len(v), dtype=np.int64), v.copy())

    def msr_scale(self, v, A):      # mono.fut:116-117 (T.* v)
        return self._mono(A["n"], A["m"], A["col_idx"].copy(), (self.np(v) * A["vals"]).astype(self.np))

    def msr_dense(self, A):
        out = np.zeros((A["n"], A["m"]), self.np)
        if A["n"] * A["m"] > 0:
            self._f("oracle_msr_dense", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]), _p(A["col_idx"]),
                                             _p(A["vals"]), _p(out))
        return out

    def msc_dense(self, A):         # mono.fut:191-192
        return np.ascontiguousarray(self.msr_dense(A).T)

    def msr_nnz(self, A):
        return int(self._f("oracle_msr_nnz", ctypes.c_int64)(ctypes.c_int64(A["n"]), _p(A["vals"])))

    def msr_coo(self, A):
        n = A["n"]
        r, c, v = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), self.np)
        k = int(self._f("oracle_msr_coo", ctypes.c_int64)(ctypes.c_int64(n), _p(A["col_idx"]), _p(A["vals"]),
                                                         _p(r), _p(c), _p(v)))
        return r[:k], c[:k], v[:k]

    def msc_coo(self, A):           # mono.fut:185-186
        r, c, v = self.msr_coo(A)
        return c, r, v

    def _msr_addsub(self, sub, A, B):
        assert A["n"] == B["n"] and A["m"] == B["m"]
        ci, va = np.zeros(A["n"], np.int64), np.zeros(A["n"], self.np)
        rc = self._f("oracle_msr_addsub")(ctypes.c_int(sub), ctypes.c_int64(A["n"]), _p(A["col_idx"]), _p(A["vals"]),
                                          _p(B["col_idx"]), _p(B["vals"]), _p(ci), _p(va))
        if rc:
            raise ValueError("oracle: column indices differ (mono.fut:141/:147 assert)")
        return self._mono(A["n"], A["m"], ci, va)

    def msr_add(self, A, B):
        return self._msr_addsub(0, A, B)

    def msr_sub(self, A, B):
        return self._msr_addsub(1, A, B)

    def msr_smvm(self, A, v):
        v = self._vals(v)
        assert len(v) == A["m"]
        y = np.zeros(A["n"], self.np)
        rc = self._f("oracle_msr_smvm")(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]), _p(A["col_idx"]),
                                        _p(A["vals"]), _p(v), _p(y))
        if rc:
            raise ValueError("oracle: index out of bounds in smvm")
        return y

    def msr_vsmm(self, v, A):
        v = self._vals(v)
        assert len(v) == A["n"]
        out = np.zeros(A["m"], self.np)
        self._f("oracle_msr_vsmm", None)(ctypes.c_int64(A["n"]), ctypes.c_int64(A["m"]), _p(v), _p(A["col_idx"]),
                                        _p(A["vals"]), _p(out))
        return out

    def msc_vsmm(self, a, B):       # mono.fut:200-201: sr.smvm (transpose b) a
        return self.msr_smvm(B, a)

    def msr_dmsmm(self, D, S):
        D = np.ascontiguousarray(np.asarray(D, dtype=self.np))
        assert D.shape[1] == S["n"]
        out = np.zeros((D.shape[0], S["m"]), self.np)
        self._f("oracle_msr_dmsmm", None)(ctypes.c_int64(D.shape[0]), ctypes.c_int64(S["n"]), ctypes.c_int64(S["m"]),
                                         _p(D), _p(S["col_idx"]), _p(S["vals"]), _p(out))
        return out


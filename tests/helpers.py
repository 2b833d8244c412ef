"""Shared test plumbing.

* `OracleCompressed` wraps oracle.oracle.Oracle in the same surface as the
  product's host mirror `sparse_amd.mk_compressed` (module names of
  compressed.fut:24-54), so one KAT runner drives both.
* `ENTRIES` restates what each `entry` of the reference's test file composes
  (compressed_test.fut:30-178) in terms of that surface.
* `load_kat()` loads the transcribed literals (tests/golden/compressed_kat.json).
"""
import json
import os

import numpy as np

from oracle.oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))

NP = {"f64": np.float64, "f32": np.float32, "i32": np.int32}


class _OSR:
    def __init__(self, O):
        self.O = O

    def zero(self, n, m): return self.O.sr_zero(n, m)
    def eye(self, n, m): return self.O.sr_eye(n, m)
    def diag(self, v): return self.O.sr_diag(v)
    def dense(self, A): return self.O.sr_dense(A)
    def scale(self, v, A): return self.O.sr_scale(v, A)
    def sparse(self, n, m, coo): return self.O.sr_sparse(n, m, *coo)
    def nnz(self, A): return self.O.sr_nnz(A)
    def coo(self, A): return self.O.sr_coo(A)
    def add(self, A, B): return self.O.sr_add(A, B)
    def sub(self, A, B): return self.O.sr_sub(A, B)
    def transpose(self, A): return A            # compressed.fut:185-186
    def smvm(self, A, v): return self.O.sr_smvm(A, v)
    def smm(self, A, B): return self.O.sr_smm(A, B)
    def to_sc(self, A): return self.O.sr_to_sc(A)


class _OSC:
    """sc.mat[n][m] = sr.mat[m][n] (compressed.fut:226): every member delegates."""
    def __init__(self, O):
        self.O = O

    def zero(self, n, m): return self.O.sr_zero(m, n)
    def eye(self, n, m): return self.O.sr_eye(m, n)
    def diag(self, v): return self.O.sr_diag(v)
    def dense(self, A): return self.O.sc_dense(A)
    def scale(self, v, A): return self.O.sr_scale(v, A)
    def sparse(self, n, m, coo): return self.O.sc_sparse(n, m, *coo)
    def nnz(self, A): return self.O.sr_nnz(A)
    def coo(self, A): return self.O.sc_coo(A)
    def add(self, A, B): return self.O.sr_add(A, B)
    def sub(self, A, B): return self.O.sr_sub(A, B)
    def transpose(self, A): return A            # compressed.fut:220-221
    def vsmm(self, a, B): return self.O.sc_vsmm(a, B)
    def to_sr(self, A): return self.O.sr_to_sc(A)


class OracleCompressed:
    def __init__(self, dtype):
        self.dtype = dtype
        self.O = Oracle(dtype)
        self.sr = _OSR(self.O)
        self.sc = _OSC(self.O)

    def smsmm(self, A, B):
        return self.O.smsmm(A, B)


class _OMSR:
    """mono sr over the oracle (mono.fut:93-163)."""
    def __init__(self, O):
        self.O = O

    def zero(self, n, m): return self.O.msr_zero(n, m)
    def eye(self, n, m): return self.O.msr_eye(n, m)
    def diag(self, v): return self.O.msr_diag(v)
    def dense(self, A): return self.O.msr_dense(A)
    def scale(self, v, A): return self.O.msr_scale(v, A)
    def sparse(self, n, m, coo): return self.O.msr_sparse(n, m, *coo)
    def nnz(self, A): return self.O.msr_nnz(A)
    def coo(self, A): return self.O.msr_coo(A)
    def add(self, A, B): return self.O.msr_add(A, B)
    def sub(self, A, B): return self.O.msr_sub(A, B)
    def transpose(self, A): return A            # mono.fut:151-152
    def smvm(self, A, v): return self.O.msr_smvm(A, v)
    def vsmm(self, v, A): return self.O.msr_vsmm(v, A)
    def dmsmm(self, D, S): return self.O.msr_dmsmm(D, S)


class _OMSC:
    """sc.mat[n][m] = sr.mat[m][n] (mono.fut:203): every member delegates (:166-201)."""
    def __init__(self, O):
        self.O = O

    def zero(self, n, m): return self.O.msr_zero(m, n)
    def eye(self, n, m): return self.O.msr_eye(m, n)
    def diag(self, v): return self.O.msr_diag(v)
    def dense(self, A): return self.O.msc_dense(A)
    def scale(self, v, A): return self.O.msr_scale(v, A)
    def sparse(self, n, m, coo): return self.O.msc_sparse(n, m, *coo)
    def nnz(self, A): return self.O.msr_nnz(A)
    def coo(self, A): return self.O.msc_coo(A)
    def add(self, A, B): return self.O.msr_add(A, B)
    def sub(self, A, B): return self.O.msr_sub(A, B)
    def transpose(self, A): return A            # mono.fut:197-198
    def vsmm(self, a, B): return self.O.msc_vsmm(a, B)


class OracleMono:
    def __init__(self, dtype):
        self.dtype = dtype
        self.O = Oracle(dtype)
        self.sr = _OMSR(self.O)
        self.sc = _OMSC(self.O)


# --- the reference's test entries, restated over the module surface ---------
def _coo(xs, ys, vs):
    return (xs, ys, vs)


ENTRIES = {
    # compressed_test.fut:30-31
    "test_csr_eye": lambda c, n, m: [c.sr.dense(c.sr.eye(n, m))],
    # :44-45
    "test_csr_sparse": lambda c, n, m, xs, ys, vs: [c.sr.dense(c.sr.sparse(n, m, _coo(xs, ys, vs)))],
    # :57-59
    "test_csr_smvm": lambda c, n, m, xs, ys, vs, v: [c.sr.smvm(c.sr.sparse(n, m, _coo(xs, ys, vs)), v)],
    # :68-69
    "test_csr_nnz": lambda c, n, m, xs, ys, vs: [c.sr.nnz(c.sr.sparse(n, m, _coo(xs, ys, vs)))],
    # :76-78
    "test_csr_coo": lambda c, n, m, xs, ys, vs: list(c.sr.coo(c.sr.sparse(n, m, _coo(xs, ys, vs)))),
    # :104-105
    "test_csc_eye": lambda c, n, m: [c.sc.dense(c.sc.eye(n, m))],
    # :118-119
    "test_csc_sparse": lambda c, n, m, xs, ys, vs: [c.sc.dense(c.sc.sparse(n, m, _coo(xs, ys, vs)))],
    # :128-129
    "test_csc_nnz": lambda c, n, m, xs, ys, vs: [c.sc.nnz(c.sc.sparse(n, m, _coo(xs, ys, vs)))],
    # :136-138
    "test_csc_coo": lambda c, n, m, xs, ys, vs: list(c.sc.coo(c.sc.sparse(n, m, _coo(xs, ys, vs)))),
    # :145-147
    "test_csr_transpose": lambda c, n, m, xs, ys, vs:
        [c.sc.dense(c.sr.transpose(c.sr.sparse(n, m, _coo(xs, ys, vs))))],
    # :154-156
    "test_csc_transpose": lambda c, n, m, xs, ys, vs:
        [c.sr.dense(c.sc.transpose(c.sc.sparse(n, m, _coo(xs, ys, vs))))],
    # :172-178
    "test_smsmm": lambda c, n, m, k, xs1, ys1, vs1, xs2, ys2, vs2:
        [c.sr.dense(c.smsmm(c.sr.sparse(n, m, _coo(xs1, ys1, vs1)),
                            c.sc.sparse(m, k, _coo(xs2, ys2, vs2))))],
}


MONO_ENTRIES = {
    # mono_test.fut:30-31
    "test_msr_eye": lambda c, n, m: [c.sr.dense(c.sr.eye(n, m))],
    # :42-43
    "test_msr_sparse": lambda c, n, m, xs, ys, vs: [c.sr.dense(c.sr.sparse(n, m, _coo(xs, ys, vs)))],
    # :55-57
    "test_msr_smvm": lambda c, n, m, xs, ys, vs, v: [c.sr.smvm(c.sr.sparse(n, m, _coo(xs, ys, vs)), v)],
    # :66-67
    "test_msr_nnz": lambda c, n, m, xs, ys, vs: [c.sr.nnz(c.sr.sparse(n, m, _coo(xs, ys, vs)))],
    # :74-76
    "test_msr_coo": lambda c, n, m, xs, ys, vs: list(c.sr.coo(c.sr.sparse(n, m, _coo(xs, ys, vs)))),
    # :87-89
    "test_msr_vsmm": lambda c, n, m, v, xs, ys, vs: [c.sr.vsmm(v, c.sr.sparse(n, m, _coo(xs, ys, vs)))],
    # :114-115
    "test_msc_eye": lambda c, n, m: [c.sc.dense(c.sc.eye(n, m))],
    # :126-127
    "test_msc_sparse": lambda c, n, m, xs, ys, vs: [c.sc.dense(c.sc.sparse(n, m, _coo(xs, ys, vs)))],
    # :136-137
    "test_msc_nnz": lambda c, n, m, xs, ys, vs: [c.sc.nnz(c.sc.sparse(n, m, _coo(xs, ys, vs)))],
    # :144-146
    "test_msc_coo": lambda c, n, m, xs, ys, vs: list(c.sc.coo(c.sc.sparse(n, m, _coo(xs, ys, vs)))),
    # :153-155
    "test_msr_transpose": lambda c, n, m, xs, ys, vs:
        [c.sc.dense(c.sr.transpose(c.sr.sparse(n, m, _coo(xs, ys, vs))))],
    # :162-164
    "test_msc_transpose": lambda c, n, m, xs, ys, vs:
        [c.sr.dense(c.sc.transpose(c.sc.sparse(n, m, _coo(xs, ys, vs))))],
}


def _decode(item, shape_hint=None):
    ty = {"i64": np.int64, "i32": np.int32}[item["type"]]
    v = item["value"]
    if isinstance(v, dict) and "empty" in v:
        return np.zeros(v["empty"], ty)
    if isinstance(v, list):
        return np.array(v, dtype=ty)
    return int(v)


def load_kat(name="compressed_kat.json"):
    doc = json.load(open(os.path.join(HERE, "golden", name)))
    out = []
    for c in doc["cases"]:
        out.append((c["entry"], c["ref"], [_decode(i) for i in c["input"]],
                    [_decode(o) for o in c["output"]]))
    return out


def run_kat(c, entry, inputs):
    return (ENTRIES.get(entry) or MONO_ENTRIES[entry])(c, *inputs)


def assert_kat_equal(got, want, ref):
    assert len(got) == len(want), ref
    for g, w in zip(got, want):
        if isinstance(w, int):
            assert int(g) == w, (ref, g, w)
        else:
            g = np.asarray(g)
            assert g.shape == w.shape, (ref, g.shape, w.shape)
            assert np.array_equal(g.astype(np.int64), w.astype(np.int64)), (ref, g, w)


# --- synthetic COO generators (host side, seeded) ----------------------------
def random_coo(n, m, nnz0, dtype, seed=42, dup_frac=0.05):
    """Uniform random triples with a controlled share of exact duplicates."""
    from sparse_amd import synth
    r = (synth.splitmix64(seed, np.arange(nnz0, dtype=np.uint64)) % np.uint64(max(n, 1))).astype(np.int64)
    c = (synth.splitmix64(seed + 1, np.arange(nnz0, dtype=np.uint64)) % np.uint64(max(m, 1))).astype(np.int64)
    ndup = int(nnz0 * dup_frac)
    if ndup and nnz0 > ndup:
        src = (synth.splitmix64(seed + 2, np.arange(ndup, dtype=np.uint64)) % np.uint64(nnz0 - ndup)).astype(np.int64)
        r[nnz0 - ndup:] = r[src]
        c[nnz0 - ndup:] = c[src]
    v = synth.uniform_values(seed + 3, nnz0, dtype)
    return r, c, v

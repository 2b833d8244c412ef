"""GPU parity for the mono format (mk_mono, mono.fut): the HIP path through the C ABI against
the CPU oracle on the same seeded inputs, and the reference's own known-answer vectors
(mono_test.fut, all i32) through the module mirror and through the generated entry names.
Bars: index arrays and i32 exact; f64 within 1e-12, f32 within 1e-5 relative.  vsmm/dmsmm add
the contributions of a column in ascending row order on both sides, so integer-valued
inputs must agree exactly."""
import ctypes

import numpy as np
import pytest

from helpers import NP, OracleMono, assert_kat_equal, load_kat, run_kat
from oracle.oracle import Oracle
from sparse_amd import synth
from test_mono_oracle import mono_coo

pytestmark = pytest.mark.gpu

KAT = load_kat("mono_kat.json")
RTOL = {"f64": 1e-12, "f32": 1e-5, "i32": 0}


@pytest.fixture(scope="module")
def ctx():
    from sparse_amd import Context
    c = Context(0)
    yield c
    c.sync()


def mk(ctx, t):
    from sparse_amd import mk_mono
    return mk_mono(t, ctx)


def close(got, want, dtype):
    got = np.asarray(got)
    assert got.shape == np.asarray(want).shape
    if dtype == "i32":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL[dtype], atol=0)


def same_record(A, want):
    ci, va = (np.asarray(a) for a in A.fields())
    assert np.array_equal(ci, want["col_idx"]) and ci.dtype == np.int64
    assert va.tobytes() == want["vals"].tobytes()


@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_kat_through_module_mirror(ctx, entry, ref, inputs, want):
    got = run_kat(mk(ctx, "i32"), entry, inputs)
    got = [g if isinstance(g, int) else np.asarray(g) for g in got]
    assert_kat_equal(got, want, ref)


@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_kat_through_generated_entry_names(ctx, entry, ref, inputs, want):
    """futhark_entry_test_ms{r,c}_*: the symbols `futhark c --library mono_test.fut` would export."""
    L, h = ctx.L, ctx.h
    f = getattr(L, "futhark_entry_" + entry)
    held, args = [], []
    for x in inputs:
        if isinstance(x, int):
            args.append(ctypes.c_int64(x))
        else:
            a = ctx.array(x, "i64" if x.dtype == np.int64 else "i32")
            held.append(a)
            args.append(a.h)
    if entry.endswith("_nnz"):
        out = ctypes.c_int64()
        assert f(h, ctypes.byref(out), *args) == 0, ctx.error()
        got = [out.value]
    elif entry.endswith("_coo"):
        o = [ctypes.c_void_p() for _ in range(3)]
        assert f(h, *[ctypes.byref(p) for p in o], *args) == 0, ctx.error()
        got = [np.asarray(ctx._out(o[0], "i64")), np.asarray(ctx._out(o[1], "i64")), np.asarray(ctx._out(o[2], "i32"))]
    else:
        o = ctypes.c_void_p()
        assert f(h, ctypes.byref(o), *args) == 0, ctx.error()
        rank = 1 if entry in ("test_msr_smvm", "test_msr_vsmm") else 2
        got = [np.asarray(ctx._out(o, "i32", rank))]
    assert_kat_equal(got, want, ref)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("shape", [(300, 170), (1, 5), (5, 1), (40000, 90000), (70000, 300)], ids=lambda s: "%dx%d" % s)
def test_surface_matches_oracle(ctx, dtype, shape):
    O, M = Oracle(dtype), mk(ctx, dtype)
    n, m = shape
    r, c, v = mono_coo(n, m, dtype, 5)
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.msr_sparse(n, m, r, c, v)
    same_record(A, Ao)                       # duplicates folded in input order: values bit-equal
    x = synth.uniform_values(7, m, dtype)
    close(M.sr.smvm(A, x), O.msr_smvm(Ao, x), dtype)
    a = synth.uniform_values(8, n, dtype)
    got, want = np.asarray(M.sr.vsmm(a, A)), O.msr_vsmm(a, Ao)
    if dtype == "i32":
        assert np.array_equal(got, want)
    else:                                     # sums of many terms: scale the tolerance by the column's magnitude
        np.testing.assert_allclose(got, want, rtol=RTOL[dtype] * 100, atol=RTOL[dtype] * 100)
    assert M.sr.nnz(A) == O.msr_nnz(Ao)
    for g, w in zip(M.sr.coo(A), O.msr_coo(Ao)):
        assert np.array_equal(np.asarray(g), w)
    same_record(M.sr.scale(3, A), O.msr_scale(3, Ao))
    same_record(M.sr.add(A, M.sr.scale(2, A)), O.msr_add(Ao, O.msr_scale(2, Ao)))
    same_record(M.sr.sub(A, A), O.msr_sub(Ao, Ao))
    if n * m <= 10 ** 6:
        assert np.array_equal(np.asarray(M.sr.dense(A)), O.msr_dense(Ao))
        D = synth.uniform_values(9, 6 * n, dtype).reshape(6, n)
        got, want = np.asarray(M.sr.dmsmm(D, A)), O.msr_dmsmm(D, Ao)
        if dtype == "i32":
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=RTOL[dtype] * 100, atol=RTOL[dtype] * 100)
    # transpose relabels, column format delegates (mono.fut:166-203)
    T = M.sr.transpose(A)
    for g, w in zip(M.sc.coo(T), O.msc_coo(Ao)):
        assert np.array_equal(np.asarray(g), w)
    close(M.sc.vsmm(x, T), O.msc_vsmm(x, Ao), dtype)
    B, Bo = M.sc.sparse(m, n, (c, r, v)), O.msc_sparse(m, n, c, r, v)
    same_record(B, Bo)
    same_record(M.sc.transpose(B), Ao)
    if n * m <= 10 ** 6:
        assert np.array_equal(np.asarray(M.sc.dense(B)), O.msc_dense(Bo))


def test_vsmm_integer_valued_floats_are_exact_and_reproducible(ctx):
    O, M = Oracle("f64"), mk(ctx, "f64")
    n, m = 200000, 37                         # ~5400 rows per column
    r, c, v = mono_coo(n, m, "f64", 21, fill=1.0, dup=False)
    v = np.round(v * 8.0)
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.msr_sparse(n, m, r, c, v)
    a = np.round(synth.uniform_values(22, n, "f64") * 16.0)
    y0 = np.asarray(M.sr.vsmm(a, A))
    assert np.array_equal(y0, O.msr_vsmm(a, Ao))
    for _ in range(3):
        assert np.array_equal(np.asarray(M.sr.vsmm(a, A)), y0)


def test_constructors_and_the_eye_quirk(ctx):
    O, M = Oracle("i32"), mk(ctx, "i32")
    for n, m in ((3, 3), (2, 5), (5, 2), (0, 0), (0, 3), (3, 0)):
        same_record(M.sr.eye(n, m), O.msr_eye(n, m))
        same_record(M.sr.zero(n, m), O.msr_zero(n, m))
        assert np.array_equal(np.asarray(M.sr.dense(M.sr.eye(n, m))), O.msr_dense(O.msr_eye(n, m)))
        assert np.array_equal(np.asarray(M.sc.dense(M.sc.eye(n, m))), O.msc_dense(O.msr_eye(m, n)))
    d = np.arange(1, 8, dtype=np.int32)
    same_record(M.sr.diag(d), O.msr_diag(d))
    # eye 5 2 stores columns 0..4 for a 2-column matrix (mono.fut:102-105): vsmm drops them, smvm fails
    from sparse_amd.compressed import SparseError
    E, Eo = M.sr.eye(5, 2), O.msr_eye(5, 2)
    ones = np.ones(5, np.int32)
    assert np.array_equal(np.asarray(M.sr.vsmm(ones, E)), O.msr_vsmm(ones, Eo))
    with pytest.raises(SparseError):
        M.sr.smvm(E, np.ones(2, np.int32))


def test_asserts_of_the_reference(ctx):
    from sparse_amd.compressed import SparseError
    M = mk(ctx, "f64")
    i64 = lambda *a: np.array(a, np.int64)   # noqa: E731
    with pytest.raises(SparseError):
        M.sr.sparse(2, 3, (i64(0, 0), i64(0, 1), np.array([1.0, 2.0])))     # two elements in one row
    with pytest.raises(SparseError):
        M.sc.sparse(2, 3, (i64(0, 1), i64(1, 1), np.array([1.0, 2.0])))     # two elements in one column
    with pytest.raises(SparseError):
        M.sr.sparse(2, 3, (i64(2), i64(0), np.array([1.0])))
    A = M.sr.sparse(2, 3, (i64(0, 0), i64(1, 1), np.array([1.0, 2.0])))     # duplicates merge first
    assert np.asarray(A.fields()[1]).tolist() == [3.0, 0.0]
    B = M.sr.sparse(2, 3, (i64(0), i64(2), np.array([5.0])))
    with pytest.raises(SparseError):
        M.sr.add(A, B)
    with pytest.raises(SparseError):
        M.sr.smvm(A, np.ones(2))
    with pytest.raises(SparseError):
        M.sr.dmsmm(np.ones((4, 3)), A)
    # the context stays usable
    assert M.sr.nnz(A) == 1

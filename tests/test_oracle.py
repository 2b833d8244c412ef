"""Pins the CPU oracle (oracle/compressed_oracle.c) before anything trusts it.

1. every known-answer vector of the reference's own test file
   (compressed_test.fut:13-178, transcribed to tests/golden/compressed_kat.json);
2. the array-level expectations derived by hand from those vectors;
3. scipy.sparse as an INDEPENDENT cross-check for what the reference does not
   pin (floats, larger shapes, SpMM, conversion).  scipy is not the reference.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import NP, OracleCompressed, assert_kat_equal, load_kat, random_coo, run_kat
from oracle import oracle as orc
from sparse_amd import synth

KAT = load_kat()


def test_kat_file_is_complete():
    # 12 entries / 36 cases
    assert len(KAT) == 36
    assert len({e for e, _, _, _ in KAT}) == 12


@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_oracle_matches_reference_kat(entry, ref, inputs, want):
    c = OracleCompressed("i32")
    assert_kat_equal(run_kat(c, entry, inputs), want, ref)


def test_derived_array_level_expectations():
    O = orc.Oracle("i32")
    # smvm case, compressed_test.fut:49-55
    A = O.sr_sparse(5, 5, [0, 0, 0, 1, 1, 2, 2, 2, 3, 4, 4], [0, 1, 3, 1, 2, 1, 2, 3, 3, 3, 4],
                    [1, 2, 11, 3, 4, 5, 6, 7, 8, 9, 10])
    assert A["row_off"].tolist() == [3, 5, 8, 9, 11]
    assert A["col_idx"].tolist() == [0, 1, 3, 1, 2, 1, 2, 3, 3, 3, 4]
    assert O.sr_smvm(A, [3, 1, 2, 6, 5]).tolist() == [71, 11, 59, 48, 104]
    # duplicate case :39-40
    A = O.sr_sparse(2, 3, [1, 0, 1], [2, 0, 2], [3, 2, 1])
    assert (A["row_off"].tolist(), A["col_idx"].tolist(), A["vals"].tolist()) == ([1, 2], [0, 2], [2, 4])
    # empty case :41-42
    A = O.sr_sparse(2, 3, [], [], [])
    assert A["row_off"].tolist() == [0, 0] and A["nnz"] == 0
    # eye 3 2 (compressed.fut:105-113)
    A = O.sr_eye(3, 2)
    assert (A["row_off"].tolist(), A["col_idx"].tolist(), A["vals"].tolist()) == ([1, 2, 2], [0, 1], [1, 1])
    # CSC of [[2,0,0],[0,0,3]]: stored record has length-m row_off
    A = O.sc_sparse(2, 3, [0, 1], [0, 2], [2, 3])
    assert (A["row_off"].tolist(), A["col_idx"].tolist(), A["vals"].tolist()) == ([1, 1, 2], [0, 1], [2, 3])


def test_out_of_range_index_is_an_error():
    O = orc.Oracle("f64")
    for r, c in (([2], [0]), ([-1], [0]), ([0], [3]), ([0], [-1])):
        with pytest.raises(ValueError):
            O.sr_sparse(2, 3, r, c, [1.0])


def test_explicit_zero_is_kept_but_not_counted():
    # merge keeps a run summing to zero (compressed.fut:76-84); nnz filters by value (:162-164)
    O = orc.Oracle("i32")
    A = O.sr_sparse(2, 2, [0, 0, 1], [1, 1, 0], [5, -5, 7])
    assert A["nnz"] == 2 and A["vals"].tolist() == [0, 7]
    assert O.sr_nnz(A) == 1


def _scipy_csr(n, m, r, c, v, dtype):
    M = sp.coo_matrix((v.astype(np.float64), (r, c)), shape=(n, m)).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    return M


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("shape", [(1, 1, 3), (7, 5, 40), (300, 200, 5000), (1000, 1000, 20000), (64, 4096, 3000)])
def test_assembly_against_scipy(dtype, shape):
    n, m, k = shape
    O = orc.Oracle(dtype)
    r, c, v = random_coo(n, m, k, dtype, seed=7 + n, dup_frac=0.2)
    A = O.sr_sparse(n, m, r, c, v)
    # scipy drops nothing here either (explicit zeros stay after sum_duplicates)
    M = sp.coo_matrix((np.ones(k), (r, c)), shape=(n, m)).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    assert np.array_equal(M.indptr[1:], A["row_off"])           # inclusive row_off == indptr[1:]
    assert np.array_equal(M.indices, A["col_idx"])
    S = _scipy_csr(n, m, r, c, v, dtype)
    if dtype == "i32":
        assert np.array_equal(S.data.astype(np.int64), A["vals"].astype(np.int64))
    else:
        np.testing.assert_allclose(A["vals"], S.data, rtol=1e-12 if dtype == "f64" else 1e-5)
    # sc.sparse is the CSR of the transpose
    B = O.sc_sparse(n, m, r, c, v)
    Mt = M.T.tocsr()
    Mt.sort_indices()
    assert np.array_equal(Mt.indptr[1:], B["row_off"]) and np.array_equal(Mt.indices, B["col_idx"])


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_products_and_conversion_against_scipy(dtype):
    n, m, k = 400, 300, 6000
    O = orc.Oracle(dtype)
    r, c, v = random_coo(n, m, k, dtype, seed=3)
    A = O.sr_sparse(n, m, r, c, v)
    S = _scipy_csr(n, m, r, c, v, dtype)
    x = synth.uniform_values(11, m, dtype)
    tol = {"f64": 1e-12, "f32": 1e-5, "i32": 0}[dtype]
    np.testing.assert_allclose(O.sr_smvm(A, x).astype(np.float64), S @ x.astype(np.float64), rtol=tol, atol=0)
    assert np.array_equal(O.sr_smvm(A, x), O.sr_smvm(A, x, threads=4))      # threaded twin is bit-identical
    B = synth.uniform_values(12, m * 8, dtype).reshape(m, 8)
    np.testing.assert_allclose(O.sr_smm(A, B).astype(np.float64), S @ B.astype(np.float64), rtol=tol, atol=0)
    # physical conversion: same matrix, column-major record
    T = O.sr_to_sc(A)
    St = S.tocsc()
    St.sort_indices()
    assert np.array_equal(St.indptr[1:], T["row_off"]) and np.array_equal(St.indices, T["col_idx"])
    np.testing.assert_allclose(T["vals"].astype(np.float64), St.data, rtol=tol, atol=0)
    assert np.array_equal(np.sort(T["vals"]), np.sort(A["vals"]))             # values only permuted, bit-exact
    a = synth.uniform_values(13, n, dtype)
    np.testing.assert_allclose(O.sc_vsmm(a, T).astype(np.float64), a.astype(np.float64) @ S, rtol=tol, atol=0)
    # coo round trip
    rr, cc, vv = O.sr_coo(A)
    A2 = O.sr_sparse(n, m, rr, cc, vv)
    for f in ("row_off", "col_idx", "vals"):
        assert np.array_equal(A[f], A2[f])
    # + and -
    r2, c2, v2 = random_coo(n, m, k // 2, dtype, seed=5)
    Bm = O.sr_sparse(n, m, r2, c2, v2)
    S2 = _scipy_csr(n, m, r2, c2, v2, dtype)
    for fn, ref in ((O.sr_add, S + S2), (O.sr_sub, S - S2)):
        R = fn(A, Bm)
        np.testing.assert_allclose(O.sr_dense(R).astype(np.float64), ref.toarray(),
                                   rtol=tol, atol=0 if dtype == "i32" else 1e-6)


def test_splitmix_twins_agree():
    idx = np.array([0, 1, 2, 12345, 2**40 + 7], dtype=np.uint64)
    got = synth.splitmix64(42, idx)
    for i, g in zip(idx, got):
        assert int(g) == orc.splitmix64(42, int(i))


def test_generators_shapes():
    g = 16
    r, c, v = synth.laplacian_2d_coo(g)
    assert len(r) == 5 * g * g - 4 * g                      # nnz = 5n - 4g
    O = orc.Oracle("f64")
    A = O.sr_sparse(g * g, g * g, r, c, v)
    y = O.sr_smvm(A, np.ones(g * g))
    # row sums of the 5-point stencil: 0 in the interior, >0 on the boundary
    gy, gx = np.divmod(np.arange(g * g), g)
    nb = (gx > 0).astype(int) + (gx < g - 1) + (gy > 0) + (gy < g - 1)
    assert np.array_equal(y, 4.0 - nb)
    p = synth.affine_shuffle(1000)
    assert np.array_equal(np.sort(p), np.arange(1000))
    r, c, v = synth.banded_coo(50, 3, "f64")
    assert len(r) == sum(50 - abs(d) for d in range(-3, 4))
    r, c, v = synth.rmat_coo(8, 4, "f32")
    assert r.max() < 256 and c.max() < 256 and len(r) == 1024

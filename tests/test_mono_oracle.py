"""Pins the mono part of the CPU oracle (oracle/mono_oracle.inc) on the reference's own
known-answer vectors (mono_test.fut:13-164, transcribed to tests/golden/mono_kat.json; all
i32) and checks it against a dense numpy formulation for what those vectors do not cover
(floats, larger shapes, dmsmm, + and -).  numpy here is an independent cross-check, not the
reference."""
import numpy as np
import pytest

from helpers import NP, OracleMono, assert_kat_equal, load_kat, run_kat
from oracle.oracle import Oracle
from sparse_amd import synth

KAT = load_kat("mono_kat.json")


def test_kat_file_is_complete():
    assert len(KAT) == 30
    assert len({e for e, _, _, _ in KAT}) == 12


@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_oracle_matches_reference_kat(entry, ref, inputs, want):
    assert_kat_equal(run_kat(OracleMono("i32"), entry, inputs), want, ref)


def mono_coo(n, m, dtype, seed, fill=0.7, dup=True):
    """One element in ~fill of the rows, random columns, optionally split into two duplicate triples."""
    i = np.arange(n, dtype=np.uint64)
    keep = synth.uniform01(seed, n) < fill
    r = np.arange(n, dtype=np.int64)[keep]
    c = (synth.splitmix64(seed + 1, i) % np.uint64(max(m, 1))).astype(np.int64)[keep]
    v = synth.uniform_values(seed + 2, n, dtype)[keep]
    if dup and len(r) > 4:
        k = len(r) // 4
        r, c, v = np.concatenate([r, r[:k]]), np.concatenate([c, c[:k]]), np.concatenate([v, v[:k]])
    p = synth.affine_shuffle(len(r))
    return r[p], c[p], v[p]


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_against_dense_numpy(dtype):
    O = Oracle(dtype)
    n, m = 300, 170
    r, c, v = mono_coo(n, m, dtype, 5)
    A = O.msr_sparse(n, m, r, c, v)
    D = np.zeros((n, m), NP[dtype])
    np.add.at(D, (r, c), v)
    tol = dict(rtol=1e-6 if dtype == "f32" else 1e-12, atol=0)
    np.testing.assert_allclose(O.msr_dense(A), D, **tol)
    x = synth.uniform_values(7, m, dtype)
    np.testing.assert_allclose(O.msr_smvm(A, x), D @ x, **tol)
    a = synth.uniform_values(8, n, dtype)
    np.testing.assert_allclose(O.msr_vsmm(a, A), a @ D, rtol=1e-5 if dtype == "f32" else 1e-12, atol=1e-6 if dtype == "f32" else 0)
    Dn = synth.uniform_values(9, 6 * n, dtype).reshape(6, n)
    np.testing.assert_allclose(O.msr_dmsmm(Dn, A), Dn @ D, rtol=1e-5 if dtype == "f32" else 1e-12, atol=1e-5 if dtype == "f32" else 0)
    assert O.msr_nnz(A) == int(np.count_nonzero(A["vals"]))
    rr, cc, vv = O.msr_coo(A)
    assert np.array_equal(rr, np.nonzero(A["vals"])[0]) and np.array_equal(cc, A["col_idx"][rr])
    S = O.msr_add(A, O.msr_scale(2, A))
    np.testing.assert_allclose(O.msr_dense(S), 3 * D, **tol)
    assert not O.msr_dense(O.msr_sub(A, A)).any()
    # column format: the stored record is that of the transpose
    r2, c2, v2 = mono_coo(m, n, dtype, 11)                 # one element per COLUMN of an n x m matrix
    B = O.msc_sparse(n, m, c2, r2, v2)
    assert (B["n"], B["m"]) == (m, n)
    D2 = np.zeros((n, m), NP[dtype])
    np.add.at(D2, (c2, r2), v2)
    np.testing.assert_allclose(O.msc_dense(B), D2, **tol)
    np.testing.assert_allclose(O.msc_vsmm(a, B), a @ D2, **tol)


def test_asserts():
    O = Oracle("i32")
    with pytest.raises(ValueError):
        O.msr_sparse(2, 3, [0, 0], [0, 1], [1, 2])          # two elements in row 0
    with pytest.raises(ValueError):
        O.msr_sparse(2, 3, [2], [0], [1])                   # row out of range
    with pytest.raises(ValueError):
        O.msr_sparse(2, 3, [0], [3], [1])                   # column out of range
    A = O.msr_sparse(2, 3, [0, 0], [1, 1], [1, 2])          # duplicates merge first: fine
    assert A["col_idx"].tolist() == [1, 0] and A["vals"].tolist() == [3, 0]
    B = O.msr_sparse(2, 3, [0], [2], [5])
    with pytest.raises(ValueError):
        O.msr_add(A, B)                                     # col_idx differ
    E = O.msr_eye(3, 2)                                     # col_idx = iota 3 although m = 2 (mono.fut:102-105)
    assert O.msr_dense(E).tolist() == [[1, 0], [0, 1], [0, 0]]
    assert O.msr_vsmm([1, 1, 1], E).tolist() == [1, 1]      # reduce_by_index ignores the out-of-range bin
    with pytest.raises(ValueError):
        O.msr_smvm(E, [1, 1])
    # explicit zero: stored, not counted, not listed by coo
    Z = O.msr_sparse(2, 2, [0, 1], [1, 0], [0, 4])
    assert O.msr_nnz(Z) == 1 and [a.tolist() for a in O.msr_coo(Z)] == [[1], [0], [4]]

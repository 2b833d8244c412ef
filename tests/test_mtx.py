"""Matrix Market loader (sparse_amd/csrc/mtx.hpp through the C ABI; host code, no GPU).
scipy.io.mmread / mmwrite are the independent cross-check; the reference has no file format, so
there is nothing of its own to pin this on."""
import io
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

from sparse_amd import synth
from sparse_amd.io import MtxError, read_mtx, write_mtx


def dense_of(n, m, coo):
    D = np.zeros((n, m))
    np.add.at(D, (coo[0], coo[1]), coo[2])
    return D


def test_general_real_with_comments_blank_lines_and_crlf(tmp_path):
    p = tmp_path / "a.mtx"
    p.write_bytes(b"%%MatrixMarket matrix coordinate real general\r\n% a comment\r\n%\r\n\r\n"
                  b"3 4 5\r\n1 1 2.5\r\n3 4 -1e-3\r\n  2 2   +7  \r\n\r\n% trailing comment\r\n3 1 1E2\r\n1 1 0.5\r\n")
    n, m, (r, c, v) = read_mtx(p)
    assert (n, m) == (3, 4)
    assert r.tolist() == [0, 2, 1, 2, 0] and c.tolist() == [0, 3, 1, 0, 0]
    assert v.tolist() == [2.5, -1e-3, 7.0, 100.0, 0.5]        # file order kept, duplicates not merged here
    assert r.dtype == np.int64 and v.dtype == np.float64


@pytest.mark.parametrize("field,sym", [("real", "general"), ("integer", "general"), ("pattern", "general"),
                                       ("real", "symmetric"), ("pattern", "symmetric"), ("real", "skew-symmetric")])
def test_against_scipy(tmp_path, field, sym):
    n = 60
    rng = np.random.default_rng(3)
    A = sp.random(n, n, density=0.08, random_state=rng, format="coo")
    if field == "integer":
        A.data = np.round(A.data * 100)
    if sym == "symmetric":
        A = sp.coo_matrix(sp.triu(A) + sp.triu(A, 1).T)
    if sym == "skew-symmetric":
        A = sp.coo_matrix(sp.triu(A, 1) - sp.triu(A, 1).T)
    p = str(tmp_path / "m.mtx")
    scipy.io.mmwrite(p, A, field=field, symmetry=sym)
    want = scipy.io.mmread(p).toarray()
    nn, mm, coo = read_mtx(p, threads=3)
    assert (nn, mm) == (n, n)
    np.testing.assert_array_equal(dense_of(nn, mm, coo), want)


def test_parallel_parse_keeps_file_order(tmp_path):
    """A body large enough to be cut into chunks: every thread count gives the same triple list."""
    k = 200000
    i = np.arange(k, dtype=np.uint64)
    r = (synth.splitmix64(1, i) % np.uint64(5000)).astype(np.int64)
    c = (synth.splitmix64(2, i) % np.uint64(7000)).astype(np.int64)
    v = synth.uniform_values(3, k, "f64")
    p = tmp_path / "big.mtx"
    write_mtx(p, 5000, 7000, (r, c, v))
    assert os.path.getsize(p) > 16 * 65536
    for threads in (1, 4, 16):
        n, m, (rr, cc, vv) = read_mtx(p, threads=threads)
        assert (n, m) == (5000, 7000)
        assert np.array_equal(rr, r) and np.array_equal(cc, c)
        assert vv.tobytes() == v.tobytes()               # 17 significant digits round-trip every f64
    # and scipy reads what we wrote
    np.testing.assert_array_equal(scipy.io.mmread(str(p)).toarray(), dense_of(5000, 7000, (r, c, v)))


def test_dtype_conversion_and_empty(tmp_path):
    p = tmp_path / "e.mtx"
    p.write_text("%%MatrixMarket matrix coordinate integer general\n4 2 0\n")
    n, m, (r, c, v) = read_mtx(p, dtype="i32")
    assert (n, m, len(r), len(c), len(v)) == (4, 2, 0, 0, 0) and v.dtype == np.int32
    p.write_text("%%MatrixMarket matrix coordinate integer general\n2 2 2\n1 2 -3\n2 1 9\n")
    _, _, (_, _, v) = read_mtx(p, dtype="i32")
    assert v.tolist() == [-3, 9] and v.dtype == np.int32
    _, _, (_, _, v) = read_mtx(p, dtype="f32")
    assert v.dtype == np.float32


@pytest.mark.parametrize("text,msg", [
    ("hello\n1 1 1\n", "banner"),
    ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", "layout"),
    ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n", "field"),
    ("%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1\n", "symmetry"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n", "out of range"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n", "out of range"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", "declares"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1\n", "malformed"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 x\n", "malformed"),
    ("%%MatrixMarket matrix coordinate real symmetric\n2 3 1\n1 1 1\n", "non-square"),
    ("%%MatrixMarket matrix coordinate real general\n% only comments\n", "size line"),
])
def test_errors(tmp_path, text, msg):
    p = tmp_path / "bad.mtx"
    p.write_text(text)
    with pytest.raises(MtxError, match=msg):
        read_mtx(p)


def test_missing_file(tmp_path):
    with pytest.raises(MtxError, match="cannot open"):
        read_mtx(tmp_path / "nope.mtx")

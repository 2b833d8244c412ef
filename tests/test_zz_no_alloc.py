"""The entry point that allocates nothing (sparse_amd_csr_*_smvm_into_noalloc, include/sparse_amd.h) against
the existing sparse_amd_csr_*_smvm_into, through the C ABI: two calls in a row, the second on a smaller
input, so whatever the first call left in the library's static scratch array is still there.

Bars as everywhere: i32 exact, f64 within 1e-12 relative, f32 within 1e-5 (a fresh record takes the launch-bound
kernel variant, in which the owner tile finishes short spills itself -- another summation order for those
rows -- while smvm_into on a large record uses the fix-up launch).  Collected last (test_zz_*): the round's
end runs pytest -x."""
import re

import numpy as np
import pytest

from helpers import NP, random_coo
from sparse_amd import synth

pytestmark = pytest.mark.gpu

RTOL = {"f64": 1e-12, "f32": 1e-5, "i32": 0}


@pytest.fixture(scope="module")
def ctx():
    from sparse_amd import Context
    c = Context(0)
    c.set_option("spmv_eager", 1)      # these tests look at the per-record SpMV forms: built at the first product, not the 16th
    yield c
    c.sync()


def held_bytes(ctx):
    """Device memory the context's pool holds and its live block count (futhark_context_report)."""
    m = re.search(r"device memory held: (\d+) bytes \(peak \d+, cached (\d+), live blocks (\d+)\)", ctx.report())
    return int(m.group(1)), int(m.group(2)), int(m.group(3))


def close(got, want, dtype):
    if dtype == "i32":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL[dtype], atol=0)


def banded(n, per, dtype, seed):
    i = np.arange(n, dtype=np.int64)
    rs = np.repeat(i, per)
    cs = np.repeat(i, per) + np.tile(np.arange(per, dtype=np.int64), n)
    return rs, cs, synth.uniform_values(seed, len(rs), dtype), n + per


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_noalloc_matches_smvm_into_twice_second_call_smaller(ctx, dtype):
    from sparse_amd import mk_compressed
    M = mk_compressed(dtype, ctx)
    # first: a large record (> 4096 tiles; rows of 7 cross tile ends all the time), second: a small random one
    rs, cs, vs, m1 = banded(700000, 7, dtype, 11)
    n1 = 700000
    A1 = M.sr.sparse(n1, m1, (rs, cs, vs))
    r2, c2, v2 = random_coo(3000, 2500, 40000, dtype, seed=12)
    A2 = M.sr.sparse(3000, 2500, (r2, c2, v2))
    x1 = ctx.array(synth.uniform_values(13, m1, dtype), dtype)
    x2 = ctx.array(synth.uniform_values(14, 2500, dtype), dtype)
    y1 = ctx.array(np.full(n1, 7, NP[dtype]), dtype)
    y2 = ctx.array(np.full(3000, 7, NP[dtype]), dtype)
    ctx.sync()
    before = held_bytes(ctx)
    M.sr.smvm_into_noalloc(y1, A1, x1)          # fresh records: nothing lazily built exists yet
    M.sr.smvm_into_noalloc(y2, A2, x2)
    ctx.sync()
    assert held_bytes(ctx) == before, "the allocation-free entry point took memory from the pool"
    got1, got2 = y1.numpy(), y2.numpy()
    w1 = ctx.array(np.zeros(n1, NP[dtype]), dtype)
    w2 = ctx.array(np.zeros(3000, NP[dtype]), dtype)
    M.sr.smvm_into(w1, A1, x1)                  # builds the 16-bit columns / row offsets of the large record
    M.sr.smvm_into(w2, A2, x2)
    close(got1, w1.numpy(), dtype)
    close(got2, w2.numpy(), dtype)
    # again, now that the large record holds its 16-bit arrays: the same kernels as smvm_into, bit for bit
    before = held_bytes(ctx)
    M.sr.smvm_into_noalloc(y1, A1, x1)
    M.sr.smvm_into_noalloc(y2, A2, x2)
    ctx.sync()
    assert held_bytes(ctx) == before
    assert np.array_equal(y1.numpy(), w1.numpy())
    assert np.array_equal(y2.numpy(), w2.numpy())


def test_noalloc_single_tile_empty_and_errors(ctx):
    from sparse_amd import mk_compressed
    from sparse_amd.compressed import SparseError
    M = mk_compressed("f64", ctx)
    r, c, v = random_coo(50, 60, 300, "f64", seed=3)          # one tile
    A = M.sr.sparse(50, 60, (r, c, v))
    x = ctx.array(synth.uniform_values(4, 60, "f64"), "f64")
    y = ctx.array(np.full(50, 7.0), "f64")
    M.sr.smvm_into_noalloc(y, A, x)
    np.testing.assert_allclose(y.numpy(), M.sr.smvm(A, x).numpy(), rtol=1e-12, atol=0)
    Z = M.sr.zero(50, 60)
    M.sr.smvm_into_noalloc(y, Z, x)
    assert np.all(y.numpy() == 0)
    with pytest.raises(SparseError):
        M.sr.smvm_into_noalloc(ctx.array(np.zeros(51), "f64"), A, x)
    with pytest.raises(SparseError):
        M.sr.smvm_into_noalloc(y, A, ctx.array(np.zeros(61), "f64"))


def test_noalloc_rejects_a_record_beyond_the_static_array(ctx):
    """The real array (4 MiB) holds the head sums of 524288 f64 tiles = 5.4e8 entries; a record that large is
    6 GB, so the limit is exercised with the test override that makes the library treat the array as 8 KiB
    (1024 tiles of f64): 1024 tiles pass, one entry more is refused with a clear error and y is left alone."""
    from sparse_amd import mk_compressed
    from sparse_amd.compressed import SparseError
    M = mk_compressed("f64", ctx)
    ctx.set_option("noalloc_scratch_bytes", 8192)
    try:
        for n, ok in ((1024 * 1024, True), (1024 * 1024 + 1, False)):
            A = M.sr.eye(n, n)
            x = ctx.array(np.arange(n, dtype=np.float64), "f64")
            y = ctx.array(np.full(n, 7.0), "f64")
            if ok:
                M.sr.smvm_into_noalloc(y, A, x)
                assert np.array_equal(y.numpy(), np.arange(n, dtype=np.float64))
            else:
                with pytest.raises(SparseError, match="static array"):
                    M.sr.smvm_into_noalloc(y, A, x)
                assert np.all(y.numpy() == 7.0)
    finally:
        ctx.set_option("noalloc_scratch_bytes", 0)


@pytest.mark.parametrize("dtype", ["f64", "i32"])
def test_noalloc_is_legal_under_graph_capture_on_a_fresh_record(dtype):
    """What the header offers the entry point for: a caller capturing its stream into a HIP graph.  The record has
    had no product yet (so nothing about it has been read back), the call sits between BeginCapture and EndCapture
    of the stream the context has adopted (torch.cuda.graph), and the graph is then replayed on two different
    operands.  y against the ORACLE.  The matrix has rows crossing tile ends, rows longer than a tile and a run of
    empty rows, i.e. every fix-up case; a call that waited for the stream or allocated would invalidate the capture."""
    import torch
    from oracle.oracle import Oracle
    from sparse_amd import Context, mk_compressed
    ctx = Context(0)
    M, O = mk_compressed(dtype, ctx), Oracle(dtype)
    n, m = 40000, 30000
    r, c, v = random_coo(n, m, 500000, dtype, seed=41)
    keep = (r < 9000) | (r >= 13000)                       # 4000 empty rows in a run
    r, c, v = r[keep], c[keep], v[keep]
    lr = np.full(3000, 17, np.int64)                        # one row of ~3000 entries: spans three tiles
    lc = np.arange(3000, dtype=np.int64) * 7 % m
    r, c = np.concatenate([r, lr]), np.concatenate([c, lc])
    v = np.concatenate([v, synth.uniform_values(42, 3000, dtype)])
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
    ctx.sync()
    tdt = {"f64": torch.float64, "i32": torch.int32}[dtype]
    x1, x2 = synth.uniform_values(43, m, dtype), synth.uniform_values(44, m, dtype)
    dev = torch.device("cuda", 0)
    xt = torch.as_tensor(x1, device=dev).contiguous()
    yt = torch.full((n,), 7, dtype=tdt, device=dev)
    s = torch.cuda.Stream(device=dev)
    ctx.set_stream(s.cuda_stream)
    xh, yh = ctx.array(xt, dtype), ctx.array(yt, dtype)     # wrapped in place: no device memory is taken
    before = held_bytes(ctx)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        M.sr.smvm_into_noalloc(yh, A, xh)
    assert held_bytes(ctx) == before
    for x in (x1, x2):
        xt.copy_(torch.as_tensor(x, device=dev))
        yt.fill_(7)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        close(yt.cpu().numpy(), O.sr_smvm(Ao, x), dtype)
    del g
    ctx.set_stream(None)
    ctx.sync()

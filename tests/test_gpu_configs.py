"""Every BASELINE.json config the C2 module (test_gpu_fullsize.py) does not already run, at its OWN size,
through the C ABI.

  configs[0]  C1  random 1e5 x 1e5, 1e6 seeded triples, f64  -- small enough for the oracle itself: assembly
                  bit-exact, smvm within 1e-12, physical CSR->CSC bit-exact (compressed.fut:154-160, :134-146,
                  :166-177 o :211-212)
  configs[2]  C3  RMAT scale 24, edge factor 18, f32 (3.0e8 triples) -- beyond the oracle's reach in seconds:
                  index arrays against torch's coalesce (independent of this library), the two-pass product
                  against the one-pass kernel within 1e-5, the i32 twin exactly
  configs[4]  C5  uniform random 1e7 x 1e7, 1e9 triples, f64, the whole record on one GPU (the 8-rank run is
                  the driver's; the 1/8 shard is in test_gpu_fullsize.py): the two-pass product against torch's own
                  index_add_ on the same triples within 1e-12
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env():
    import torch
    from sparse_amd import Context
    ctx = Context(0)
    ctx.set_option("spmv_eager", 1)      # these tests look at the per-record SpMV forms: built at the first product, not the 16th
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield torch, ctx
    torch.cuda.synchronize()
    ctx.clear_caches()
    ctx.close()
    torch.cuda.empty_cache()


def test_c1_random_1e5_f64_against_the_oracle(env):
    torch, ctx = env
    from oracle.oracle import Oracle
    from sparse_amd import mk_compressed, synth
    n = m = 100000
    r, c, v = synth.random_uniform_coo(n, m, 1000000, "f64", seed=42)      # BASELINE configs[0], SURVEY 8d
    O, M = Oracle("f64"), mk_compressed("f64", ctx)
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
    assert Ao["nnz"] < 1000000, "the config is meant to hold duplicate coordinates"
    assert A.stored_dims() == (n, m, Ao["nnz"])
    ro, ci, va = (np.asarray(f) for f in A.fields())
    assert np.array_equal(ro, Ao["row_off"]) and np.array_equal(ci, Ao["col_idx"])
    assert np.array_equal(va, Ao["vals"])                                   # merged left to right: bit-equal
    x = synth.uniform_values(7, m, "f64")
    want = O.sr_smvm(Ao, x)
    for algo in (0, 2, 3):
        ctx.set_option("spmv_algo", algo)
        try:
            np.testing.assert_allclose(M.sr.smvm(A, x).numpy(), want, rtol=1e-12, atol=0)
        finally:
            ctx.set_option("spmv_algo", 0)
    T, To = M.sr.to_sc(A), O.sr_to_sc(Ao)
    ro, ci, va = (np.asarray(f) for f in T.fields())
    assert np.array_equal(ro, To["row_off"]) and np.array_equal(ci, To["col_idx"]) and np.array_equal(va, To["vals"])
    # vsmm on the converted record = smvm of the transpose (compressed.fut:223-224)
    a = synth.uniform_values(8, n, "f64")
    np.testing.assert_allclose(M.sc.vsmm(a, T).numpy(), O.sr_smvm(To, a), rtol=1e-12, atol=0)
    # sparse x dense panel at this size, against the column-wise definition
    B = synth.uniform_values(9, m * 8, "f64").reshape(m, 8)
    np.testing.assert_allclose(M.sr.smm(A, B).numpy(), O.sr_smm(Ao, B), rtol=1e-12, atol=0)


def test_c3_rmat_scale_24_f32(env):
    torch, ctx = env
    from sparse_amd import mk_compressed, synth
    dev = "cuda:0"
    n = 1 << 24
    r, c, v = synth.rmat_coo_device(24, 18, dev, "f32")
    assert r.numel() == 301989888
    M = mk_compressed("f32", ctx)
    A = M.sr.sparse(n, n, (r, c, v))
    nnz = A.stored_dims()[2]
    # ---- index arrays against torch's coalesce (sort + unique of the same coordinates; no library code)
    key = torch.unique(r * n + c)                       # sorted, duplicates removed
    assert key.numel() == nnz
    ro, ci, va = A.fields()
    assert torch.equal(torch.as_tensor(ci, device=dev), key % n)
    rows = key // n
    del key
    want_ro = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    del rows
    assert torch.equal(torch.as_tensor(ro, device=dev), want_ro)
    del want_ro, ro, ci
    # merged values: sum of each coordinate's duplicates; f32 left-to-right vs torch's own order
    tot = float(v.double().sum())
    assert abs(float(torch.as_tensor(va, device=dev).double().sum()) - tot) <= 1e-6 * tot
    del va
    torch.cuda.empty_cache()
    # ---- the product: auto mode takes the two-pass path on this record
    x = torch.rand(n, device=dev, dtype=torch.float32) + 0.5
    y = torch.as_tensor(M.sr.smvm(A, x), device=dev).clone()
    assert ctx.get_option("spmv_last_path") == 3, "the two-pass path was not chosen"
    # again: the order in which the LDS atomics of pass 2 land is not fixed, so the last bit of an f32 sum may
    # differ between two runs (the f64 accumulator absorbs the rest)
    y_again = torch.as_tensor(M.sr.smvm(A, x), device=dev)
    assert float(((y_again - y).abs() / y.abs().clamp_min(1e-30)).max()) <= 2e-7
    del A, y_again
    ctx.clear_caches()
    ctx.set_option("spmv_pb", -1)
    ctx.set_option("spmv_reorder", -1)
    try:
        A1 = M.sr.sparse(n, n, (r, c, v))
        y1 = torch.as_tensor(M.sr.smvm(A1, x), device=dev).clone()
        assert ctx.get_option("spmv_last_path") == 0
        del A1
        Mi = mk_compressed("i32", ctx)
        vi = (v * 8).to(torch.int32)
        xi = (x * 8).to(torch.int32)
        Ai = Mi.sr.sparse(n, n, (r, c, vi))
        yi1 = torch.as_tensor(Mi.sr.smvm(Ai, xi), device=dev).clone()
        del Ai
    finally:
        ctx.set_option("spmv_pb", 0)
        ctx.set_option("spmv_reorder", 0)
    rel = ((y - y1).abs() / y1.abs().clamp_min(1e-30)).max()
    assert float(rel) <= 1e-5, float(rel)
    # an independent reference for the product as well: torch's COO scatter-add in f64
    want = torch.zeros(n, device=dev, dtype=torch.float64).index_add_(0, r, v.double() * x.double()[c])
    nz = want > 0
    assert float(((y.double() - want).abs()[nz] / want[nz]).max()) <= 1e-5
    del want, nz, y, y1
    ctx.clear_caches()
    Ai = Mi.sr.sparse(n, n, (r, c, vi))
    yi = torch.as_tensor(Mi.sr.smvm(Ai, xi), device=dev)
    assert ctx.get_option("spmv_last_path") == 3
    assert torch.equal(yi, yi1)                         # integers: exact whatever the path


def test_c5_random_1e9_f64_whole_record_on_one_gpu(env):
    torch, ctx = env
    from sparse_amd import mk_compressed, synth
    dev = "cuda:0"
    n = m = 10**7
    r, c, v = synth.random_uniform_coo_device(n, m, 10**9, dev)
    M = mk_compressed("f64", ctx)
    A = M.sr.sparse(n, m, (r, c, v))
    nnz = A.stored_dims()[2]
    assert 10**9 - 10**7 < nnz <= 10**9                  # ~5e3 duplicate coordinates expected (nnz0^2 / 2nm)
    ctx.clear_caches()
    x = torch.rand(m, device=dev, dtype=torch.float64) + 0.5
    y = torch.as_tensor(M.sr.smvm(A, x), device=dev).clone()
    assert ctx.get_option("spmv_last_path") == 3, "the two-pass path was not chosen"
    y_again = torch.as_tensor(M.sr.smvm(A, x), device=dev)      # reproducible up to the order of the LDS atomics
    assert float(((y_again - y).abs() / y.abs()).max()) <= 1e-13    # ~100 terms per row, a few ulp of reordering
    del y_again
    # row_off against torch's histogram of the same row ids (duplicates counted once: compare the totals
    # and monotonicity, then the exact histogram on the rows of the merged record)
    ro = torch.as_tensor(A.fields()[0], device=dev)
    assert int(ro[-1]) == nnz and bool(torch.all(ro[1:] >= ro[:-1]))
    del ro
    want = torch.zeros(n, device=dev, dtype=torch.float64)
    step = 250 * 10**6                                   # in slices: the temporary v * x[c] stays at 2 GB
    for lo in range(0, 10**9, step):
        want.index_add_(0, r[lo:lo + step], v[lo:lo + step] * x[c[lo:lo + step]])
    assert float(((y - want).abs() / want.abs()).max()) <= 1e-12

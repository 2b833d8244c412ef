"""Full-size (BASELINE.json configs[1]: 5-point Laplacian, 4096^2 grid, 8.4e7 entries)
checks through properties that do not need the oracle at that size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    from sparse_amd import Context, mk_compressed, synth
    ctx = Context(0)
    ctx.set_option("spmv_eager", 1)      # these tests look at the per-record SpMV forms: built at the first product, not the 16th
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    M = mk_compressed("f64", ctx)
    g = 4096
    r, c, v = synth.laplacian_2d_coo_device(g, "cuda:0")
    A = M.sr.sparse(g * g, g * g, (r, c, v))
    del r, c, v
    torch.cuda.empty_cache()
    return torch, ctx, M, A, g


def test_c2_structure(setup):
    torch, ctx, M, A, g = setup
    n = g * g
    assert A.stored_dims() == (n, n, 5 * n - 4 * g)
    ro, ci, va = A.fields()
    ro_t = torch.as_tensor(ro, device="cuda:0")
    ci_t = torch.as_tensor(ci, device="cuda:0")
    # row_off: inclusive prefix sum of the stencil's row lengths
    i = torch.arange(n, device="cuda:0")
    gx, gy = i % g, i // g
    lens = 1 + (gx > 0).long() + (gx < g - 1).long() + (gy > 0).long() + (gy < g - 1).long()
    assert torch.equal(ro_t, torch.cumsum(lens, 0))
    # columns strictly increasing inside every row
    starts = torch.zeros(5 * n - 4 * g, dtype=torch.bool, device="cuda:0")
    starts[0] = True
    starts[ro_t[:-1]] = True
    inc = ci_t[1:] > ci_t[:-1]
    assert bool(torch.all(inc | starts[1:]))
    # the matrix is symmetric: its CSC record equals its CSR record, bit for bit
    T = M.sr.to_sc(A)
    ro2, ci2, va2 = T.fields()
    assert torch.equal(torch.as_tensor(ro2, device="cuda:0"), ro_t)
    assert torch.equal(torch.as_tensor(ci2, device="cuda:0"), ci_t)
    assert torch.equal(torch.as_tensor(va2, device="cuda:0"), torch.as_tensor(va, device="cuda:0"))


def test_c2_smvm_equals_stencil(setup):
    torch, ctx, M, A, g = setup
    from sparse_amd import synth
    n = g * g
    x = 1.0 + (torch.arange(n, device="cuda:0") % 7).double() / 8.0   # exactly representable: sums are exact
    y = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0")
    assert torch.equal(y, synth.laplacian_2d_apply_device(g, x))
    # linearity on exactly representable data: A(2x + e_k) = 2 A x + A e_k
    e = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    e[12345] = 1.0
    lhs = torch.as_tensor(M.sr.smvm(A, 2 * x + e), device="cuda:0")
    rhs = 2 * y + torch.as_tensor(M.sr.smvm(A, e), device="cuda:0")
    assert torch.equal(lhs, rhs)
    # every SpMV algorithm agrees at full size
    for algo in (2, 3):
        ctx.set_option("spmv_algo", algo)
        try:
            assert torch.equal(torch.as_tensor(M.sr.smvm(A, x), device="cuda:0"), y)
        finally:
            ctx.set_option("spmv_algo", 0)


def test_c2_coo_roundtrip_is_idempotent(setup):
    torch, ctx, M, A, g = setup
    n = g * g
    r, c, v = M.sr.coo(A)
    A2 = M.sr.sparse(n, n, (r, c, v))
    for a, b in zip(A.fields(), A2.fields()):
        assert torch.equal(torch.as_tensor(a, device="cuda:0"), torch.as_tensor(b, device="cuda:0"))


def test_dist_csr_single_rank_on_gpu():
    """The row-sharded wrapper with a world of one: torch CUDA tensors in, the same kernel underneath."""
    import torch
    from sparse_amd import Context, mk_compressed, synth
    from sparse_amd.distributed import DistCsr
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    M = mk_compressed("f64", ctx)
    g = 512
    n = g * g
    r, c, v = synth.laplacian_2d_coo_device(g, "cuda:0")
    D = DistCsr.from_local_coo(M, None, n, n, 0, n, r, c, v)
    assert D.mode == "none" and D.x_len == n
    D.x_own.copy_(synth.laplacian_x_device(0, n, "cuda:0"))
    y = torch.as_tensor(D.smvm(), device="cuda:0")
    assert torch.equal(y, synth.laplacian_2d_apply_rows(g, g, 0, n, "cuda:0"))


# ---- the other BASELINE configs at full size, through properties that need no oracle run at that size
def _ctx():
    import torch
    from sparse_amd import Context
    ctx = Context(0)
    ctx.set_option("spmv_eager", 1)      # these tests look at the per-record SpMV forms: built at the first product, not the 16th
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return torch, ctx


def test_c4_smm_matrix_cores_against_the_entry_path_and_linearity():
    """configs[3]: banded n = 2^20, 27 entries per row, k = 64, f64.  The dense-block (MFMA) tiles against the
    same record multiplied entry by entry (spmm_algo 3): different kernels, different summation order, results
    within 1e-12 relative (all values positive).  Column j of the product equals smvm with column j of the
    panel (the definition), and the product is linear in the panel."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    M = mk_compressed("f64", ctx)
    n, k = 1 << 20, 64
    r, c, v = synth.banded_coo_device(n, 13, "cuda:0")
    A = M.sr.sparse(n, n, (r, c, v))
    del r, c, v
    B = torch.rand(n, k, device="cuda:0", dtype=torch.float64) + 0.5
    C = torch.as_tensor(M.sr.smm(A, B), device="cuda:0").clone()
    ctx.set_option("spmm_algo", 3)
    try:
        C3 = torch.as_tensor(M.sr.smm(A, B), device="cuda:0").clone()
    finally:
        ctx.set_option("spmm_algo", 0)
    assert float(((C - C3).abs() / C3.abs()).max()) <= 1e-12
    for j in (0, 17, 63):
        yj = torch.as_tensor(M.sr.smvm(A, B[:, j].contiguous()), device="cuda:0")
        assert float(((C[:, j] - yj).abs() / yj.abs()).max()) <= 1e-12
    C2 = torch.as_tensor(M.sr.smm(A, 2.0 * B), device="cuda:0")
    assert torch.equal(C2, 2.0 * C)          # scaling by a power of two is exact in every partial sum


def test_c5_row_shard_two_pass_and_column_blocks_against_the_plain_kernel():
    """configs[4], what one of 8 ranks holds: uniform random 1.25e6 x 1e7, 1.25e8 triples, f64.  Auto mode
    picks the two-pass product for a wide record of flat column popularity; with spmv_pb -1 it falls back to the
    column-blocked product, with spmv_cblk -1 as well to the plain streaming kernel.  Three kernels, three
    summation orders: within 1e-12 relative of each other and of torch's own scatter-add; the column-blocked
    one is reproducible bit for bit, the two-pass one up to the order its LDS atomics land in."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    M = mk_compressed("f64", ctx)
    n, m = 1250000, 10**7
    r, c, v = synth.random_uniform_coo_device(n, m, 125 * 10**6, "cuda:0")
    x = torch.rand(m, device="cuda:0", dtype=torch.float64) + 0.5
    A = M.sr.sparse(n, m, (r, c, v))
    y = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0").clone()
    assert ctx.get_option("spmv_last_path") == 3, "the two-pass path was not chosen"
    y_again = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0")
    assert float(((y - y_again).abs() / y.abs()).max()) <= 1e-13   # ~100 terms per row, a few ulp of reordering
    del A
    ctx.set_option("spmv_pb", -1)
    try:
        A1 = M.sr.sparse(n, m, (r, c, v))       # a fresh record: the choice is made at a record's first product
        y1 = torch.as_tensor(M.sr.smvm(A1, x), device="cuda:0").clone()
        assert ctx.get_option("spmv_last_path") == 2, "column blocks were not chosen"
        assert torch.equal(torch.as_tensor(M.sr.smvm(A1, x), device="cuda:0"), y1)
        del A1
        ctx.set_option("spmv_cblk", -1)
        A2 = M.sr.sparse(n, m, (r, c, v))
        y2 = torch.as_tensor(M.sr.smvm(A2, x), device="cuda:0")
        assert ctx.get_option("spmv_last_path") == 0
    finally:
        ctx.set_option("spmv_cblk", 0)
        ctx.set_option("spmv_pb", 0)
    assert float(((y - y2).abs() / y2.abs()).max()) <= 1e-12
    assert float(((y1 - y2).abs() / y2.abs()).max()) <= 1e-12
    # and against torch's own COO path on the same triples (independent of this library)
    want = torch.zeros(n, device="cuda:0", dtype=torch.float64).index_add_(0, r, v * x[c])
    assert float(((y - want).abs() / want.abs()).max()) <= 1e-12


def test_c5_stress_shard_against_torch_scatter_add():
    """SURVEY 8(d)'s stress configuration (n = m = 1e8, about 10 entries per row) as what one of 8 ranks holds
    after row sharding: uniform random 1.25e7 x 1e8, 1.25e8 triples, f64 -- x is 800 MB, every gather a miss.
    Assembly: index arrays equal to torch's own sort of the triples; product: within 1e-12 of torch's scatter-add,
    whatever path the library picks (the test records it)."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    M = mk_compressed("f64", ctx)
    n, m = 12500000, 10**8
    r, c, v = synth.random_uniform_coo_device(n, m, 125 * 10**6, "cuda:0")
    x = torch.rand(m, device="cuda:0", dtype=torch.float64) + 0.5
    A = M.sr.sparse(n, m, (r, c, v))
    want = torch.zeros(n, device="cuda:0", dtype=torch.float64).index_add_(0, r, v * x[c])
    key = torch.unique(r * m + c)                   # sorted, duplicates merged: the record's (row, column) order
    ro, ci, va = A.fields()
    assert A.stored_dims()[2] == key.numel()
    assert torch.equal(torch.as_tensor(ci, device="cuda:0"), key % m)
    assert torch.equal(torch.as_tensor(ro, device="cuda:0"), torch.cumsum(torch.bincount(key // m, minlength=n), 0))
    del key, ro, ci, va, r, c, v
    y = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0")
    path = ctx.get_option("spmv_last_path")
    assert path in (0, 1, 2, 3)
    nz = want != 0
    assert float(((y - want).abs()[nz] / want.abs()[nz]).max()) <= 1e-12, "path %d" % path
    assert bool((y[~nz] == 0).all())


def test_c3_hot_column_renumbering_against_the_plain_kernel():
    """configs[2] at scale 22 (same generator and skew as scale 24, a sixteenth of the entries so the module
    stays within a few seconds): the renumbered product is bit-identical to the plain one."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    M = mk_compressed("f32", ctx)
    n = 1 << 22
    r, c, v = synth.rmat_coo_device(22, 18, "cuda:0", "f32")
    x = torch.rand(n, device="cuda:0", dtype=torch.float32) + 0.5
    ys = []
    for mode in (1, -1):
        ctx.set_option("spmv_reorder", mode)
        try:
            A = M.sr.sparse(n, n, (r, c, v))
            ys.append(torch.as_tensor(M.sr.smvm(A, x), device="cuda:0").clone())
        finally:
            ctx.set_option("spmv_reorder", 0)
    assert torch.equal(ys[0], ys[1])


def test_c3_two_pass_product_is_chosen_and_within_tolerance():
    """configs[2] at scale 22: a large, wide, skewed f32 record takes the two-pass product (x and y slices in
    LDS) by itself; against the plain one-pass kernel on a fresh record within the f32 bar (1e-5 relative: the
    two-pass sum is accumulated in f64 in another order), and for integer values exactly."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    n = 1 << 22
    r, c, v = synth.rmat_coo_device(22, 18, "cuda:0", "f32")
    x = torch.rand(n, device="cuda:0", dtype=torch.float32) + 0.5
    M = mk_compressed("f32", ctx)
    A = M.sr.sparse(n, n, (r, c, v))
    y = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0").clone()
    assert ctx.get_option("spmv_last_path") == 3, "the two-pass path was not chosen"
    y_again = torch.as_tensor(M.sr.smvm(A, 2 * x), device="cuda:0").clone()      # cached pieces, another x
    ctx.set_option("spmv_pb", -1)
    ctx.set_option("spmv_reorder", -1)
    try:
        A2 = M.sr.sparse(n, n, (r, c, v))
        y2 = torch.as_tensor(M.sr.smvm(A2, x), device="cuda:0").clone()
        assert ctx.get_option("spmv_last_path") == 0
        assert float(((y - y2).abs() / y2.abs().clamp_min(1e-30)).max()) <= 1e-5
        assert float(((y_again - 2 * y2).abs() / y2.abs().clamp_min(1e-30)).max()) <= 2e-5
        Mi = mk_compressed("i32", ctx)
        vi = (v * 8).to(torch.int32)
        xi = (x * 8).to(torch.int32)
        yi2 = torch.as_tensor(Mi.sr.smvm(Mi.sr.sparse(n, n, (r, c, vi)), xi), device="cuda:0").clone()
    finally:
        ctx.set_option("spmv_pb", 0)
        ctx.set_option("spmv_reorder", 0)
    yi = torch.as_tensor(Mi.sr.smvm(Mi.sr.sparse(n, n, (r, c, vi)), xi), device="cuda:0")
    assert ctx.get_option("spmv_last_path") == 3
    assert torch.equal(yi, yi2)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_large_assembly_reports_an_out_of_range_index_and_recovers(dtype):
    """6e6 triples (large enough for the path that samples the triples before packing them, counts digits while
    packing and sorts the top key bits first): one row index equal to n, then one negative column, must fail with
    the reference's bounds message whichever kernels ran; the same context then assembles the clean input, and that
    record equals the one the plain LSD sort builds."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed
    from sparse_amd.compressed import SparseError
    M = mk_compressed(dtype, ctx)
    n = m = 3 * 10**6
    k = 6 * 10**6
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    r = torch.randint(0, n, (k,), generator=g, device="cuda:0", dtype=torch.int64)
    c = torch.randint(0, m, (k,), generator=g, device="cuda:0", dtype=torch.int64)
    v = (torch.rand(k, generator=g, device="cuda:0", dtype=torch.float64) + 0.5).to(
        torch.float64 if dtype == "f64" else torch.float32)
    for arr, pos, bad in ((r, 4321, n), (c, k - 7, -1)):
        keep = int(arr[pos])
        arr[pos] = bad
        with pytest.raises(SparseError, match="out of bounds"):
            M.sr.sparse(n, m, (r, c, v))
        arr[pos] = keep
    A = M.sr.sparse(n, m, (r, c, v))
    assert ctx.get_option("sort_last_msd") > 0
    ctx.set_option("sort_msd", -1)
    try:
        B = M.sr.sparse(n, m, (r, c, v))
    finally:
        ctx.set_option("sort_msd", 0)
    for fa, fb in zip(A.fields(), B.fields()):
        assert torch.equal(torch.as_tensor(fa, device="cuda:0"), torch.as_tensor(fb, device="cuda:0"))


def test_skewed_f64_record_takes_the_folded_two_pass_product():
    """RMAT scale 22 with 8-byte values: the lines of columns are sorted into the blocks by popularity and pass 1 adds
    up what a row holds in a piece, which is what makes two passes pay for 8-byte values (scale 24: 2.45 -> 1.46 ms).
    Against the plain one-pass kernel on a fresh record within the f64 bar, against torch's scatter-add, and the
    dealt, unfolded layout forced on the same record for comparison."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed, synth
    n = 1 << 22
    r, c, v = synth.rmat_coo_device(22, 18, "cuda:0", "f64")
    x = torch.rand(n, device="cuda:0", dtype=torch.float64) + 0.5
    M = mk_compressed("f64", ctx)
    A = M.sr.sparse(n, n, (r, c, v))
    y = torch.as_tensor(M.sr.smvm(A, x), device="cuda:0").clone()
    assert ctx.get_option("spmv_last_path") == 3 and ctx.get_option("spmv_last_pb_fold") == 1
    assert ctx.get_option("spmv_last_pb_outputs") < 0.7 * A.stored_dims()[2]     # most products are summed in pass 1
    want = torch.zeros(n, device="cuda:0", dtype=torch.float64).index_add_(0, r, v * x[c])
    nz = want > 0
    assert float(((y - want).abs()[nz] / want[nz]).max()) <= 1e-12
    del A
    ctx.clear_caches()
    for opts, path in (({"spmv_pb": -1, "spmv_reorder": -1}, 0), ({"spmv_pb": 2, "spmv_pb_fold": -1}, 3)):
        for k, val in opts.items():
            ctx.set_option(k, val)
        try:
            A2 = M.sr.sparse(n, n, (r, c, v))
            y2 = torch.as_tensor(M.sr.smvm(A2, x), device="cuda:0").clone()
            assert ctx.get_option("spmv_last_path") == path
            if path == 3:
                assert ctx.get_option("spmv_last_pb_fold") == 0
            assert float(((y - y2).abs() / y2.abs().clamp_min(1e-300)).max()) <= 1e-12
            del A2
        finally:
            for k in opts:
                ctx.set_option(k, 0)
        ctx.clear_caches()


@pytest.mark.parametrize("dtype,sorted_input", [("f32", False), ("i32", False), ("f64", False), ("f64", True)])
def test_assembly_second_sort_equals_the_lsd_sort_at_scale(dtype, sorted_input):
    """3e7 uniform random triples on 4e6 x 4e6 (44 key bits, ~0.01 % duplicate coordinates): in auto mode the
    assembly takes the top key bits through global passes and sorts, folds and writes the rest from LDS
    (radix_sort_msd.hpp); with sort_msd -1 it is the plain LSD sort + merge_emit, which the oracle pins at small
    sizes.  The two records must be equal bit for bit -- 4-byte values, 8-byte values carried with their keys, and
    (sorted input) 8-byte values fetched through their source index."""
    torch, ctx = _ctx()
    from sparse_amd import mk_compressed
    M = mk_compressed(dtype, ctx)
    n = m = 4 * 10**6
    k = 3 * 10**7
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    r = torch.randint(0, n, (k,), device="cuda:0", generator=g)
    c = torch.randint(0, m, (k,), device="cuda:0", generator=g)
    if dtype == "i32":
        v = torch.randint(-9, 10, (k,), device="cuda:0", generator=g, dtype=torch.int32)
    else:
        v = torch.rand(k, device="cuda:0", generator=g, dtype=torch.float32 if dtype == "f32" else torch.float64) - 0.5
    if sorted_input:
        o = torch.argsort(r * m + c, stable=True)
        r, c, v = r[o].contiguous(), c[o].contiguous(), v[o].contiguous()
        del o
    A = M.sr.sparse(n, m, (r, c, v))
    assert ctx.get_option("sort_last_msd") in (16, 24), "the second sort was not chosen"
    ctx.set_option("sort_msd", -1)
    try:
        B = M.sr.sparse(n, m, (r, c, v))
        assert ctx.get_option("sort_last_msd") == 0
    finally:
        ctx.set_option("sort_msd", 0)
    assert A.stored_dims() == B.stored_dims() and A.stored_dims()[2] < k
    for fa, fb in zip(A.fields(), B.fields()):
        ta, tb = torch.as_tensor(fa, device="cuda:0"), torch.as_tensor(fb, device="cuda:0")
        if ta.dtype.is_floating_point:
            ta, tb = ta.view(torch.int32 if ta.element_size() == 4 else torch.int64), \
                tb.view(torch.int32 if tb.element_size() == 4 else torch.int64)
        assert torch.equal(ta, tb)

"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs, plus the reference's own known-answer vectors.

Bars (BASELINE.json north_star): index arrays bit-exact; i32 exact; f64 values
within 1e-12 relative, f32 within 1e-5 relative.  Assembly values are in fact
asserted bit-exact: the stable radix sort keeps duplicates in input order and
runs are folded left to right, exactly what the oracle does.
"""
import ctypes

import numpy as np
import pytest

from helpers import NP, OracleCompressed, assert_kat_equal, load_kat, random_coo, run_kat
from oracle.oracle import Oracle
from sparse_amd import synth

pytestmark = pytest.mark.gpu

KAT = load_kat()
RTOL = {"f64": 1e-12, "f32": 1e-5, "i32": 0}


@pytest.fixture(scope="module")
def ctx():
    from sparse_amd import Context
    c = Context(0)
    c.set_option("spmv_eager", 1)      # these tests look at the per-record SpMV forms: built at the first product, not the 16th
    yield c
    c.sync()


def mk(ctx, t):
    from sparse_amd import mk_compressed
    return mk_compressed(t, ctx)


def fields(A):
    ro, ci, va = A.fields()
    return ro.numpy(), ci.numpy(), va.numpy()


def assert_same_record(A, want, exact_vals=True, dtype="f64"):
    ro, ci, va = fields(A)
    n, m, nnz = A.stored_dims()
    assert (n, m, nnz) == (want["n"], want["m"], want["nnz"])
    assert np.array_equal(ro, want["row_off"]), "row_off differs"
    assert np.array_equal(ci, want["col_idx"]), "col_idx differs"
    if exact_vals:
        assert np.array_equal(va, want["vals"]), "vals differ"
    else:
        np.testing.assert_allclose(va, want["vals"], rtol=RTOL[dtype], atol=0)


def close(got, want, dtype):
    got = np.asarray(got)
    if dtype == "i32":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL[dtype], atol=0)


# ----------------------------------------------------------- reference KATs
@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_kat_through_module_mirror(ctx, entry, ref, inputs, want):
    assert_kat_equal(run_kat(mk(ctx, "i32"), entry, inputs), want, ref)


_RAW_OUT = {
    "test_csr_eye": ["i32_2d"], "test_csc_eye": ["i32_2d"], "test_csr_sparse": ["i32_2d"],
    "test_csc_sparse": ["i32_2d"], "test_csr_transpose": ["i32_2d"], "test_csc_transpose": ["i32_2d"],
    "test_smsmm": ["i32_2d"], "test_csr_smvm": ["i32_1d"], "test_csr_nnz": ["i64"], "test_csc_nnz": ["i64"],
    "test_csr_coo": ["i64_1d", "i64_1d", "i32_1d"], "test_csc_coo": ["i64_1d", "i64_1d", "i32_1d"],
}


@pytest.mark.parametrize("entry,ref,inputs,want", KAT, ids=[k[0] + "@" + k[1].split(":")[1] for k in KAT])
def test_kat_through_generated_entry_names(ctx, entry, ref, inputs, want):
    """Calls futhark_entry_<name> exactly as a harness for the reference's generated library would."""
    from sparse_amd.compressed import DeviceArray
    L = ctx.L
    args, keep = [], []
    for x in inputs:
        if isinstance(x, int):
            args.append(ctypes.c_int64(x))
        else:
            d = ctx.array(x, "i64" if x.dtype == np.int64 else "i32")
            keep.append(d)
            args.append(d.h)
    outs = []
    for kind in _RAW_OUT[entry]:
        outs.append(ctypes.c_int64() if kind == "i64" else ctypes.c_void_p())
    rc = getattr(L, "futhark_entry_" + entry)(ctx.h, *[ctypes.byref(o) for o in outs], *args)
    ctx.check(rc)
    got = []
    for kind, o in zip(_RAW_OUT[entry], outs):
        if kind == "i64":
            got.append(o.value)
        else:
            t, r = kind.split("_")
            got.append(DeviceArray(ctx, o.value, t, int(r[0])).numpy())
    assert_kat_equal(got, want, ref)


# ------------------------------------------------------------ assembly parity
SHAPES = [(1, 1, 5), (7, 5, 40), (300, 200, 5000), (1000, 1000, 20000), (64, 4096, 30000), (5000, 3, 9000),
          (100000, 100000, 300000)]


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d_%d" % s)
def test_sparse_matches_oracle(ctx, dtype, shape):
    n, m, k = shape
    r, c, v = random_coo(n, m, k, dtype, seed=n + k, dup_frac=0.2)
    O = Oracle(dtype)
    M = mk(ctx, dtype)
    assert_same_record(M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v))
    assert_same_record(M.sc.sparse(n, m, (r, c, v)), O.sc_sparse(n, m, r, c, v))


@pytest.mark.parametrize("dtype", ["f64", "i32"])
def test_sparse_edge_cases(ctx, dtype):
    O, M = Oracle(dtype), mk(ctx, dtype)
    e = np.zeros(0, np.int64)
    ev = np.zeros(0, NP[dtype])
    for n, m in ((2, 3), (0, 0), (0, 4), (4, 0), (1, 1)):
        assert_same_record(M.sr.sparse(n, m, (e, e, ev)), O.sr_sparse(n, m, e, e, ev))
    # every triple the same coordinate: one run, folded left to right (kLongRun = 4096 pairs is where a workgroup takes over)
    k = 4096
    r, c = np.full(k, 3, np.int64), np.full(k, 1, np.int64)
    v = synth.uniform_values(1, k, dtype)
    assert_same_record(M.sr.sparse(5, 2, (r, c, v)), O.sr_sparse(5, 2, r, c, v))
    # all rows empty but the last; single column
    r, c = np.full(100, 999, np.int64), np.zeros(100, np.int64)
    assert_same_record(M.sr.sparse(1000, 1, (r, c, v[:100])), O.sr_sparse(1000, 1, r, c, v[:100]))
    # already sorted / reverse sorted input
    r, c, v = random_coo(50, 60, 700, dtype, seed=9, dup_frac=0.0)
    o = np.lexsort((c, r))
    for order in (o, o[::-1]):
        assert_same_record(M.sr.sparse(50, 60, (r[order], c[order], v[order])),
                           O.sr_sparse(50, 60, r[order], c[order], v[order]))
    # wide key: more than 32 key bits, and a column range past int32 (col_idx stays i64 on device)
    n, m = 1 << 20, 1 << 33
    r = (synth.splitmix64(5, np.arange(3000, dtype=np.uint64)) % np.uint64(n)).astype(np.int64)
    c = (synth.splitmix64(6, np.arange(3000, dtype=np.uint64)) % np.uint64(m)).astype(np.int64)
    v = synth.uniform_values(7, 3000, dtype)
    assert_same_record(M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v))


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("algo", [0, 1], ids=["fold_emit", "scan_merge_emit"])
def test_sparse_long_runs_of_one_coordinate_are_summed_by_a_workgroup(ctx, dtype, algo):
    """merge_coo (compressed.fut:76-84) when one coordinate repeats more than kLongRun = 4096 times: the run leaves
    the one-thread left-to-right fold and is summed by a workgroup (fold_long_runs_kernel).  Index arrays and every
    short run stay bit-equal to the oracle; the long runs are exact for i32 and within the bars for floats (the
    reference's segmented_reduce fixes no order).  Runs of 4097 (just over), 20 000 and 300 000, one of them at the
    very end of the input, next to ordinary entries; both emission paths (sort_algo 1 = scan + merge_emit)."""
    n, m = 3000, 2000
    r, c, v = random_coo(n, m, 60000, dtype, seed=71, dup_frac=0.1)
    runs = [(5, 7, 4097), (1234, 0, 20000), (n - 1, m - 1, 300000)]
    rr = np.concatenate([r] + [np.full(k, a, np.int64) for a, b, k in runs])
    cc = np.concatenate([c] + [np.full(k, b, np.int64) for a, b, k in runs])
    vv = np.concatenate([v] + [synth.uniform_values(72 + i, k, dtype) for i, (a, b, k) in enumerate(runs)])
    perm = np.argsort(synth.splitmix64(73, np.arange(len(rr), dtype=np.uint64)), kind="stable")   # shuffled input
    rr, cc, vv = rr[perm], cc[perm], vv[perm]
    O, M = Oracle(dtype), mk(ctx, dtype)
    want = O.sr_sparse(n, m, rr, cc, vv)
    ctx.set_option("sort_algo", algo)
    try:
        A = M.sr.sparse(n, m, (rr, cc, vv))
        if algo == 0:
            assert ctx.get_option("sort_last_long_runs") == 3
    finally:
        ctx.set_option("sort_algo", 0)
    ro, ci, va = fields(A)
    assert np.array_equal(ro, want["row_off"]) and np.array_equal(ci, want["col_idx"])
    pos = [int(want["row_off"][a - 1] if a else 0) + int(np.searchsorted(ci[(want["row_off"][a - 1] if a else 0):want["row_off"][a]], b))
           for a, b, k in runs]
    short = np.ones(len(va), bool)
    short[pos] = False
    assert np.array_equal(va[short], want["vals"][short]), "a short run lost its left-to-right sum"
    if dtype == "i32":
        assert np.array_equal(va[pos], want["vals"][pos])
    else:
        # the workgroup's sum against the exactly rounded one (math.fsum); the oracle's one-at-a-time f32 sum is the
        # less accurate of the two on 300 000 terms, so it is the oracle that gets the looser look here
        import math
        for p_, (a, b, k) in zip(pos, runs):
            sel = (rr == a) & (cc == b)
            exact = math.fsum(vv[sel].astype(np.float64))
            assert abs(float(va[p_]) - exact) <= RTOL[dtype] * abs(exact), (dtype, k, float(va[p_]), exact)
        np.testing.assert_allclose(va[pos], want["vals"][pos], rtol=1e-12 if dtype == "f64" else 2e-3, atol=0)


@pytest.mark.parametrize("dtype", ["f64", "i32"])
def test_sparse_one_coordinate_repeated_ten_million_times(ctx, dtype):
    """The bounded fold at the size the one-thread loop could not survive: 10^7 triples on ONE coordinate plus a
    few others.  i32 exact (wrapping sums commute); f64 within 1e-12 of the oracle's left-to-right sum."""
    import time
    k = 10 ** 7
    rr = np.concatenate([np.full(k, 2, np.int64), np.array([0, 4, 2], np.int64)])
    cc = np.concatenate([np.full(k, 1, np.int64), np.array([0, 2, 0], np.int64)])
    vv = synth.uniform_values(81, k + 3, dtype)
    O, M = Oracle(dtype), mk(ctx, dtype)
    want = O.sr_sparse(5, 3, rr, cc, vv)
    dr, dc, dv = ctx.array(rr, "i64"), ctx.array(cc, "i64"), ctx.array(vv, dtype)
    M.sr.sparse(5, 3, (dr, dc, dv))                              # warm: pool and code objects
    ctx.sync()
    t0 = time.perf_counter()
    A = M.sr.sparse(5, 3, (dr, dc, dv))
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3
    assert ctx.get_option("sort_last_long_runs") == 1
    assert_same_record(A, want, exact_vals=(dtype == "i32"), dtype=dtype)
    assert ms < 50, "assembly of 1e7 equal coordinates took %.1f ms" % ms


@pytest.mark.parametrize("dtype,indirect,pairs", [("f64", 1, 0), ("f64", -1, 0), ("f64", -1, -1), ("f32", 0, 0), ("i32", 0, 0)])
@pytest.mark.parametrize("top,chunk", [(4, 0), (8, 700), (12, 0), (16, 2048), (20, 300), (32, 1000)])
def test_sparse_top_bits_by_global_passes_rest_in_lds(ctx, dtype, indirect, pairs, top, chunk):
    """radix_sort_msd.hpp, forced (option sort_msd = number of top key bits) on inputs the oracle can follow:
    buckets of thousands of triples down to single ones, chunks cut at bucket boundaries, one or two ordinal
    passes, and buckets that do not fit a workgroup (top = 4: sixteen buckets of ~19 K): sorted again without it.
    8-byte values that travel with their keys do so side by side with them (one array of pairs, option sort_pairs 0)
    or in an array of their own (-1)."""
    n, m, k = 100000, 100000, 300000                      # 34 key bits
    r, c, v = random_coo(n, m, k, dtype, seed=top * 31 + chunk, dup_frac=0.2)
    O, M = Oracle(dtype), mk(ctx, dtype)
    want = O.sr_sparse(n, m, r, c, v)
    ctx.set_option("sort_msd", top)
    ctx.set_option("sort_msd_chunk", chunk)
    ctx.set_option("sort_indirect", indirect)
    ctx.set_option("sort_pairs", pairs)
    try:
        assert_same_record(M.sr.sparse(n, m, (r, c, v)), want)
        assert ctx.get_option("sort_last_msd") == (-1 if top == 4 else top)
        # sorted and reverse-sorted input, and one coordinate repeated through a whole chunk
        o = np.lexsort((c, r))
        for order in (o, o[::-1]):
            assert_same_record(M.sr.sparse(n, m, (r[order], c[order], v[order])),
                               O.sr_sparse(n, m, r[order], c[order], v[order]))
        r2, c2 = r.copy(), c.copy()
        r2[1000:4000], c2[1000:4000] = 77, 5
        assert_same_record(M.sr.sparse(n, m, (r2, c2, v)), O.sr_sparse(n, m, r2, c2, v))
        # ten triples per coordinate: the count-sized output arrays are cut down to the merged length
        r4, c4 = r % 173, (c % 173) * 577
        A4 = M.sr.sparse(n, m, (r4, c4, v))
        assert_same_record(A4, O.sr_sparse(n, m, r4, c4, v))
        assert A4.stored_dims()[2] <= 173 * 173
        # a column range past int32: 50 key bits
        n3, m3 = 1 << 17, 1 << 33
        c3 = (synth.splitmix64(6, np.arange(k, dtype=np.uint64)) % np.uint64(m3)).astype(np.int64)
        assert_same_record(M.sr.sparse(n3, m3, (r, c3, v)), O.sr_sparse(n3, m3, r, c3, v))
    finally:
        ctx.set_option("sort_msd", 0)
        ctx.set_option("sort_msd_chunk", 0)
        ctx.set_option("sort_indirect", 0)
        ctx.set_option("sort_pairs", 0)


@pytest.mark.parametrize("top", [0, 8, 16])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_sparse_row_ends_with_and_without_empty_rows(ctx, dtype, top):
    """The emission kernels count the rows that hold something; when that is every row, the max-scan that fills the
    empty rows' ends in is skipped.  Every row occupied (skipped), one row empty in the middle / at the start / at the
    end (not skipped) -- with duplicates, so that the number of triples says nothing -- after the LSD sort (top = 0)
    and through the in-LDS emission (top = 8: buckets of whole rows, counted; top = 16: rows cut across buckets, not
    counted, always scanned)."""
    n, m = 5000, 3000
    O, M = Oracle(dtype), mk(ctx, dtype)
    base_r = np.arange(n, dtype=np.int64)
    extra_r = (synth.splitmix64(3, np.arange(20000, dtype=np.uint64)) % np.uint64(n)).astype(np.int64)
    ctx.set_option("sort_msd", top)
    try:
        for hole in (None, 2500, 0, n - 1):
            r = np.concatenate([base_r, extra_r, extra_r[:3000]])            # the last 3000 repeat coordinates
            c = (synth.splitmix64(4, np.arange(len(r), dtype=np.uint64)) % np.uint64(m)).astype(np.int64)
            c[-3000:] = c[n:n + 3000]
            if hole is not None:
                keep = r != hole
                r, c = r[keep], c[keep]
            v = synth.uniform_values(5, len(r), dtype)
            A = M.sr.sparse(n, m, (r, c, v))
            assert_same_record(A, O.sr_sparse(n, m, r, c, v))
    finally:
        ctx.set_option("sort_msd", 0)


def test_sparse_is_sorted_again_without_look_back_after_a_timed_out_wait(ctx):
    """A look-back wait that runs into its limit (GPU shared or preempted for seconds) no longer fails the assembly:
    it is repeated with histogram + scan passes, which wait for nothing.  Injected: option sort_inject."""
    n, m, k = 100000, 100000, 200000
    r, c, v = random_coo(n, m, k, "f64", seed=5, dup_frac=0.1)
    O, M = Oracle("f64"), mk(ctx, "f64")
    ctx.set_option("sort_inject", 1)
    try:
        assert_same_record(M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v))
        assert ctx.get_option("sort_last_msd") == -1
    finally:
        ctx.set_option("sort_inject", 0)


def test_sparse_rejects_out_of_range(ctx):
    from sparse_amd.compressed import SparseError
    M = mk(ctx, "f64")
    for r, c in (([2], [0]), ([-1], [0]), ([0], [3]), ([0], [-1])):
        with pytest.raises(SparseError):
            M.sr.sparse(2, 3, (np.array(r, np.int64), np.array(c, np.int64), np.array([1.0])))
    with pytest.raises(SparseError):
        M.sr.sparse(2, 3, (np.array([0, 1], np.int64), np.array([0], np.int64), np.array([1.0])))
    # the context stays usable after an error
    A = M.sr.sparse(2, 3, (np.array([0], np.int64), np.array([1], np.int64), np.array([2.0])))
    assert M.sr.nnz(A) == 1


# ---------------------------------------------------------------- SpMV parity
def skewed_coo(n, m, dtype, seed):
    """A few very long rows, many short ones, empty stretches (power-law-like)."""
    rs, cs = [], []
    rs.append(np.full(m, 5, np.int64)); cs.append(np.arange(m, dtype=np.int64))             # dense row
    rs.append(np.full(m // 2, n - 2, np.int64)); cs.append(np.arange(0, m - 1, 2, dtype=np.int64)[:m // 2])
    k = 4 * n
    i = np.arange(k, dtype=np.uint64)
    u = synth.uniform01(seed, k)
    rs.append((n // 3 + (u ** 4) * (n // 2)).astype(np.int64)); cs.append((synth.splitmix64(seed + 1, i) % np.uint64(m)).astype(np.int64))
    r, c = np.concatenate(rs), np.concatenate(cs)
    return r, c, synth.uniform_values(seed + 2, len(r), dtype)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("algo", [1, 2, 3])
def test_smvm_matches_oracle(ctx, dtype, algo):
    O, M = Oracle(dtype), mk(ctx, dtype)
    ctx.set_option("spmv_algo", algo)
    try:
        cases = [random_coo(1000, 1200, 9000, dtype, seed=1), random_coo(20000, 20000, 200000, dtype, seed=2),
                 skewed_coo(3000, 9000, dtype, 3), random_coo(1, 5000, 4000, dtype, seed=4),
                 random_coo(5000, 1, 300, dtype, seed=5)]
        dims = [(1000, 1200), (20000, 20000), (3000, 9000), (1, 5000), (5000, 1)]
        for (n, m), (r, c, v) in zip(dims, cases):
            A = M.sr.sparse(n, m, (r, c, v))
            Ao = O.sr_sparse(n, m, r, c, v)
            x = synth.uniform_values(17, m, dtype)
            close(M.sr.smvm(A, x), O.sr_smvm(Ao, x), dtype)
            # vsmm on the relabelled record: a B = (B^T) a
            a = synth.uniform_values(18, n, dtype)
            Bt = M.sc.sparse(n, m, (r, c, v))
            close(M.sc.vsmm(a, Bt), O.sc_vsmm(a, O.sc_sparse(n, m, r, c, v)), dtype)
    finally:
        ctx.set_option("spmv_algo", 0)


def test_smvm_laplacian_and_tile_boundaries(ctx):
    """Rows that straddle the 2048-entry tiles, rows longer than several tiles, empty rows between."""
    O, M = Oracle("f64"), mk(ctx, "f64")
    g = 96
    r, c, v = synth.laplacian_2d_coo(g)
    p = synth.affine_shuffle(len(r))
    A = M.sr.sparse(g * g, g * g, (r[p], c[p], v[p]))
    x = 1.0 + (np.arange(g * g) % 7) / 8.0
    assert np.array_equal(M.sr.smvm(A, x).numpy(), O.sr_smvm(O.sr_sparse(g * g, g * g, r, c, v), x))
    # rows of length 2047, 2049, 1, 0, 6000, 0, 0, 4096, 3
    lens = [2047, 2049, 1, 0, 6000, 0, 0, 4096, 3]
    m = 7000
    rs = np.concatenate([np.full(L, i, np.int64) for i, L in enumerate(lens)])
    cs = np.concatenate([np.arange(L, dtype=np.int64) for L in lens])
    vs = synth.uniform_values(3, len(rs), "f64")
    A = M.sr.sparse(len(lens), m, (rs, cs, vs))
    x = synth.uniform_values(4, m, "f64")
    close(M.sr.smvm(A, x), O.sr_smvm(O.sr_sparse(len(lens), m, rs, cs, vs), x), "f64")
    # integer twin of the same structure must be exact
    Oi, Mi = Oracle("i32"), mk(ctx, "i32")
    vi = synth.uniform_values(3, len(rs), "i32")
    xi = synth.uniform_values(4, m, "i32")
    assert np.array_equal(Mi.sr.smvm(Mi.sr.sparse(len(lens), m, (rs, cs, vi)), xi).numpy(),
                          Oi.sr_smvm(Oi.sr_sparse(len(lens), m, rs, cs, vi), xi))


@pytest.mark.parametrize("lanes", [0, 1, 4, 64])
def test_smvm_rows_crossing_tile_ends(ctx, lanes):
    """Rows that run 1, 63, 64, 65, 200, 1023 ... 3000 entries past the end of the 1024-entry tile they start
    in (head fragments + fix-up in tile order), at every lanes-per-row setting.  i32: exact."""
    Oi, Mi = Oracle("i32"), mk(ctx, "i32")
    lens = []
    for spill in (1, 2, 63, 64, 65, 66, 200, 1023, 1024, 1025, 3000, 0, 5):
        # pad to 64 short of a tile boundary, then one row crossing it by `spill`
        used = sum(lens) % 1024
        pad = (1024 - 64 - used) % 1024
        lens += [7] * (pad // 7) + [pad % 7] + [64 + spill, 0, 3]
    m = 4200
    cs = np.concatenate([np.sort(np.unique((np.arange(L, dtype=np.int64) * 13 + i) % m)) for i, L in enumerate(lens)])
    rs = np.concatenate([np.full(L, i, np.int64) for i, L in enumerate(lens)])
    assert len(rs) == len(cs)
    vi = synth.uniform_values(5, len(rs), "i32")
    xi = synth.uniform_values(6, m, "i32")
    want = Oi.sr_smvm(Oi.sr_sparse(len(lens), m, rs, cs, vi), xi)
    ctx.set_option("spmv_lanes", lanes)
    try:
        A = Mi.sr.sparse(len(lens), m, (rs, cs, vi))
        assert np.array_equal(Mi.sr.smvm(A, xi).numpy(), want)
        # the same structure with every row cut to <= 100 entries
        keep = np.concatenate([np.arange(L) < min(L, 100) for L in np.bincount(rs, minlength=len(lens))])
        B = Mi.sr.sparse(len(lens), m, (rs[keep], cs[keep], vi[keep]))
        assert np.array_equal(Mi.sr.smvm(B, xi).numpy(),
                              Oi.sr_smvm(Oi.sr_sparse(len(lens), m, rs[keep], cs[keep], vi[keep]), xi))
    finally:
        ctx.set_option("spmv_lanes", 0)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_smvm_hot_column_renumbering_is_bit_identical(ctx, dtype):
    """spmv_reorder=1 forces the popularity renumbering of the columns (normally kept for large power-law
    records); entry order is untouched, so y must equal the plain path bit for bit, and the oracle."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    n, m = 3000, 9000
    r, c, v = skewed_coo(n, m, dtype, 31)
    c = (c.astype(np.float64) ** 3 / float(m) ** 2).astype(np.int64)      # pile the columns up near 0
    x = synth.uniform_values(5, m, dtype)
    ys = []
    for mode in (-1, 1):
        ctx.set_option("spmv_reorder", mode)
        try:
            A = M.sr.sparse(n, m, (r, c, v))
            ys.append(M.sr.smvm(A, x).numpy())
            ys.append(M.sr.smvm(A, x).numpy())                            # second call: cached renumbering
            x2 = synth.uniform_values(6, m, dtype)                        # and a different x
            close(M.sr.smvm(A, x2), O.sr_smvm(O.sr_sparse(n, m, r, c, v), x2), dtype)
        finally:
            ctx.set_option("spmv_reorder", 0)
    assert all(y.tobytes() == ys[0].tobytes() for y in ys[1:])
    close(ys[0], O.sr_smvm(O.sr_sparse(n, m, r, c, v), x), dtype)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("width", [512, 4096])
def test_smvm_column_blocks(ctx, dtype, width):
    """spmv_cblk=W forces the column-blocked product (normally kept for wide matrices with flat column
    popularity): the record is split into blocks of W columns and y accumulates block by block.  Against the
    oracle; i32 exact; reproducible; second product on the cached split with another x; a matrix whose first
    block is empty; rows longer than a tile, empty rows and runs of them."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    ctx.set_option("spmv_cblk", width)
    try:
        for n, m, gen in ((3000, 9000, lambda: skewed_coo(3000, 9000, dtype, 8)),
                          (20000, 5000, lambda: random_coo(20000, 5000, 90000, dtype, seed=91, dup_frac=0.05)),
                          (700, 6000, lambda: tuple(a if i != 1 else a + 1500 for i, a in
                                                    enumerate(random_coo(700, 4500, 20000, dtype, seed=92))))):
            r, c, v = gen()
            A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
            x = synth.uniform_values(5, m, dtype)
            y0 = M.sr.smvm(A, x).numpy()
            close(y0, O.sr_smvm(Ao, x), dtype)
            assert M.sr.smvm(A, x).numpy().tobytes() == y0.tobytes()
            x2 = synth.uniform_values(6, m, dtype)
            close(M.sr.smvm(A, x2), O.sr_smvm(Ao, x2), dtype)
            assert_same_record(A, Ao)                       # the record itself is untouched by the split
    finally:
        ctx.set_option("spmv_cblk", 0)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("fold", [(-1, 0), (1, -1), (1, 1)], ids=["plain", "folded", "folded_sorted"])
@pytest.mark.parametrize("geom", [(0, 0, 0), (64, 2, 2048), (700, 5, 4096)], ids=["one_piece", "tiny", "mid"])
def test_smvm_two_pass_panels(ctx, dtype, geom, fold):
    """spmv_pb=2 forces the two-pass product (normally kept for large, wide, skewed records of 4-byte values): entries regrouped
    into (row panel) x (column block) pieces, products streamed out, then added per panel.  The overrides
    shrink panels and blocks so that small records have many of both.  Against the oracle within the dtype's
    bar (the summation order differs from the oracle's), i32 exact; a second x on the cached pieces; a very
    long row (longer than a panel's entry budget), empty rows, runs of them; the record itself untouched.
    fold = 1: the layout in which pass 1 adds up the products of one row inside one piece before writing them
    (runs of 1 .. thousands of entries, cut at every 32nd position), -1: every product written; deal = 1: the
    column lines sorted into the blocks by popularity (blocks of very different sizes) instead of dealt round them."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    rows, lines, entries = geom
    fold, deal = fold
    ctx.set_option("spmv_pb", 2)
    ctx.set_option("spmv_pb_fold", fold)
    ctx.set_option("spmv_pb_deal", deal)
    ctx.set_option("spmv_pb_rows", rows)
    ctx.set_option("spmv_pb_lines", lines)
    ctx.set_option("spmv_pb_entries", entries)
    try:
        for n, m, gen in ((3000, 9000, lambda: skewed_coo(3000, 9000, dtype, 8)),
                          (20000, 5000, lambda: random_coo(20000, 5000, 90000, dtype, seed=91, dup_frac=0.05)),
                          (1, 7000, lambda: random_coo(1, 7000, 6000, dtype, seed=93)),
                          (6000, 1, lambda: random_coo(6000, 1, 900, dtype, seed=94)),
                          (700, 6000, lambda: tuple(a if i != 1 else a + 1500 for i, a in
                                                    enumerate(random_coo(700, 4500, 20000, dtype, seed=92))))):
            r, c, v = gen()
            A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
            x = synth.uniform_values(5, m, dtype)
            close(M.sr.smvm(A, x), O.sr_smvm(Ao, x), dtype)
            assert ctx.get_option("spmv_last_path") == 3 and ctx.get_option("spmv_last_pb_fold") == (fold > 0)
            x2 = synth.uniform_values(6, m, dtype)
            close(M.sr.smvm(A, x2), O.sr_smvm(Ao, x2), dtype)
            ctx.set_option("spmv_pb_xcopy", 1)          # x copied into block order first (what an x that is not
            close(M.sr.smvm(A, x2), O.sr_smvm(Ao, x2), dtype)   # 16-byte aligned gets) instead of gathered by pass 1
            ctx.set_option("spmv_pb_xcopy", 0)
            # an x that really is not 16-byte aligned (a view one element into a device buffer): pass 1's wide
            # loads cannot gather from it, the library copies it into block order by itself
            import torch
            tdt = {"f64": torch.float64, "f32": torch.float32, "i32": torch.int32}[dtype]
            buf = torch.zeros(m + 1, dtype=tdt, device="cuda")
            buf[1:] = torch.as_tensor(np.asarray(x2), dtype=tdt, device="cuda")
            xv = buf[1:]
            assert xv.data_ptr() % 16 != 0
            close(M.sr.smvm(A, ctx.array(xv, dtype)), O.sr_smvm(Ao, x2), dtype)
            assert_same_record(A, Ao)
    finally:
        for k in ("spmv_pb", "spmv_pb_rows", "spmv_pb_lines", "spmv_pb_entries", "spmv_pb_fold", "spmv_pb_deal", "spmv_pb_xcopy"):
            ctx.set_option(k, 0)


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_smvm_16_bit_columns_and_row_offsets_on_a_large_banded_record(ctx, dtype):
    """Records of more than 4096 tiles read 16-bit tile-relative row offsets, and 16-bit tile-relative columns
    when every tile spans fewer than 65536 columns (banded / stencil).  Against the oracle; bit-identical to the
    32-bit-column kernel (same entries in the same order); one far-away entry makes a tile too wide and the
    record keeps its 32-bit columns; empty rows, a row longer than a tile, rows crossing tile ends."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    n, hb = 300000, 8
    i = np.arange(n, dtype=np.int64)
    rs = np.repeat(i, 2 * hb + 1)
    cs = rs + np.tile(np.arange(-hb, hb + 1, dtype=np.int64), n)
    keep = (cs >= 0) & (cs < n) & (rs % 1000 != 7)                       # every 1000th row empty
    rs, cs = rs[keep], cs[keep]
    long_r = np.full(3000, 123456, np.int64)                             # one row longer than two tiles
    long_c = 123456 - 1500 + np.arange(3000, dtype=np.int64)
    rs, cs = np.concatenate([rs, long_r]), np.concatenate([cs, long_c])
    vs = synth.uniform_values(31, len(rs), dtype)
    x = synth.uniform_values(32, n, dtype)
    A, Ao = M.sr.sparse(n, n, (rs, cs, vs)), O.sr_sparse(n, n, rs, cs, vs)
    assert A.stored_dims()[2] > 4096 * 1024
    y = M.sr.smvm(A, x).numpy()
    assert ctx.get_option("spmv_last_col_bytes") == 2
    close(y, O.sr_smvm(Ao, x), dtype)
    ctx.set_option("spmv_col16", -1)
    try:
        A2 = M.sr.sparse(n, n, (rs, cs, vs))
        y2 = M.sr.smvm(A2, x).numpy()
        assert ctx.get_option("spmv_last_col_bytes") == 4
    finally:
        ctx.set_option("spmv_col16", 0)
    assert y.tobytes() == y2.tobytes()
    # one entry far from the band: its tile spans more than 65535 columns
    rs3, cs3 = np.append(rs, 10), np.append(cs, n - 1)
    vs3 = np.append(vs, vs[0])
    A3, Ao3 = M.sr.sparse(n, n, (rs3, cs3, vs3)), O.sr_sparse(n, n, rs3, cs3, vs3)
    y3 = M.sr.smvm(A3, x).numpy()
    assert ctx.get_option("spmv_last_col_bytes") == 4
    close(y3, O.sr_smvm(Ao3, x), dtype)


def test_smvm_large_record_whose_entries_fill_whole_tiles_with_trailing_empty_rows(ctx):
    """Entry count an exact multiple of the 1024-entry tile, empty rows after the last entry (their start equals
    the entry count: the one case in which a tile-relative row offset is 1024), empty rows in front, on a
    record large enough for the 16-bit offsets and columns."""
    O, M = Oracle("f64"), mk(ctx, "f64")
    n0, per = 4200 * 128, 8                         # 4200 tiles exactly
    lead, trail = 3, 100
    i = np.arange(n0, dtype=np.int64)
    rs = np.repeat(i + lead, per)
    cs = np.repeat(i, per) + np.tile(np.arange(per, dtype=np.int64), n0)
    n, m = n0 + lead + trail, n0 + per
    vs = synth.uniform_values(41, len(rs), "f64")
    x = synth.uniform_values(42, m, "f64")
    A, Ao = M.sr.sparse(n, m, (rs, cs, vs)), O.sr_sparse(n, m, rs, cs, vs)
    assert A.stored_dims()[2] == 4200 * 1024
    y = M.sr.smvm(A, x).numpy()
    assert ctx.get_option("spmv_last_col_bytes") == 2
    close(y, O.sr_smvm(Ao, x), "f64")
    assert np.all(y[:lead] == 0) and np.all(y[-trail:] == 0)


@pytest.mark.parametrize("lanes", [0, 8])
def test_smvm_long_runs_of_empty_rows(ctx, lanes):
    """Hundreds of thousands of empty rows before, between and after the stored ones: the tile that holds a
    run's position gets every row of it (the many-empty-rows path of spmv_stream), together with short rows,
    a row longer than a tile and rows crossing tile ends.  i32 exact, f64 within tolerance."""
    n, m = 700000, 5000
    rows_with = np.concatenate([np.arange(150000, 150040), [300000], np.arange(300001, 302000, 7),
                                [650000, 699999]]).astype(np.int64)
    lens = np.full(len(rows_with), 5, np.int64)
    lens[3] = 1500                       # spans two tiles
    lens[40] = 3000                      # row 300000: longer than a tile
    lens[-1] = 900
    rs = np.repeat(rows_with, lens)
    cs = np.concatenate([np.sort(synth.affine_shuffle(m, seed=int(i))[:L]) for i, L in enumerate(lens)]).astype(np.int64)
    ctx.set_option("spmv_lanes", lanes)
    try:
        for dtype in ("i32", "f64"):
            O, M = Oracle(dtype), mk(ctx, dtype)
            v = synth.uniform_values(8, len(rs), dtype)
            x = synth.uniform_values(9, m, dtype)
            A = M.sr.sparse(n, m, (rs, cs, v))
            Ao = O.sr_sparse(n, m, rs, cs, v)
            close(M.sr.smvm(A, x), O.sr_smvm(Ao, x), dtype)
            if lanes == 0:                                   # the same record through the panel product
                B = synth.uniform_values(10, m * 8, dtype).reshape(m, 8)
                close(M.sr.smm(A, B), O.sr_smm(Ao, B), dtype)
    finally:
        ctx.set_option("spmv_lanes", 0)


@pytest.mark.parametrize("tiles", [8, 4200])
def test_smvm_and_smm_exactly_2048_rows_start_in_every_tile(ctx, tiles):
    """One entry in every odd row: 2048 rows start in every 1024-entry tile, and with the candidate head row in
    front of them a tile's block visits 2049 -- one past the limit of the row-driven path, so every tile but
    the first takes the entry-driven path that writes no zeros.  The host must then have cleared the result
    (the plan statistic and the kernels' row count share one definition).  Results go into storage that held
    sevens; 8 tiles = the launch-bound variant, 4200 = the large-record variant.  Against the oracle."""
    n = tiles * 2048
    rs = np.arange(1, n, 2, dtype=np.int64)
    cs = (rs * 7) % 1000
    m = 1000
    for dtype in ("f64", "i32"):
        O, M = Oracle(dtype), mk(ctx, dtype)
        v = synth.uniform_values(61, len(rs), dtype)
        x = synth.uniform_values(62, m, dtype)
        A, Ao = M.sr.sparse(n, m, (rs, cs, v)), O.sr_sparse(n, m, rs, cs, v)
        assert A.stored_dims()[2] == tiles * 1024
        y = ctx.array(np.full(n, 7, NP[dtype]), dtype)
        M.sr.smvm_into(y, A, x)
        want = O.sr_smvm(Ao, x)
        close(y.numpy(), want, dtype)
        assert np.all(y.numpy()[0::2] == 0)
        k = 4
        B = synth.uniform_values(63, m * k, dtype).reshape(m, k)
        junk = ctx.array(np.full((n, k), 7, NP[dtype]), dtype, 2)   # the pool hands this block to the product
        junk.free()
        C = M.sr.smm(A, B).numpy()
        close(C, O.sr_smm(Ao, B), dtype)
        assert np.all(C[0::2] == 0)


def test_smvm_empty_and_shape_errors(ctx):
    from sparse_amd.compressed import SparseError
    M = mk(ctx, "f64")
    Z = M.sr.zero(4, 3)
    assert np.array_equal(M.sr.smvm(Z, np.ones(3)).numpy(), np.zeros(4))
    assert M.sr.smvm(M.sr.zero(0, 3), np.ones(3)).numpy().shape == (0,)
    with pytest.raises(SparseError):
        M.sr.smvm(Z, np.ones(4))


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_smvm_into_caller_storage(ctx, dtype):
    """sparse_amd_csr_*_smvm_into: same bits as the allocating entry; with row indices it adds
    into the named elements only."""
    from sparse_amd.compressed import SparseError
    M = mk(ctx, dtype)
    n, m = 3000, 2500
    r, c, v = random_coo(n, m, 40000, dtype, seed=21)
    A = M.sr.sparse(n, m, (r, c, v))
    x = synth.uniform_values(3, m, dtype)
    want = M.sr.smvm(A, x).numpy()
    y = ctx.array(np.full(n, 7, NP[dtype]), dtype)
    M.sr.smvm_into(y, A, x)
    assert np.array_equal(y.numpy(), want)
    # y[rows[i]] += (A x)[i] into a longer vector
    base = synth.uniform_values(4, 2 * n, dtype)
    rows = np.random.default_rng(5).permutation(2 * n)[:n].astype(np.int64)
    big = ctx.array(base.copy(), dtype)
    M.sr.smvm_into(big, A, x, rows)
    expect = base.copy()
    expect[rows] = (expect[rows] + want).astype(NP[dtype])
    close(big.numpy(), expect, dtype)      # small records take a one-launch kernel that sums a row in another order
    untouched = np.ones(2 * n, bool)
    untouched[rows] = False
    assert np.array_equal(big.numpy()[untouched], base[untouched])
    with pytest.raises(SparseError):
        M.sr.smvm_into(ctx.array(np.zeros(n + 1, NP[dtype]), dtype), A, x)
    with pytest.raises(SparseError):
        M.sr.smvm_into(big, A, x, rows[:-1])


@pytest.mark.parametrize("kind", ["banded", "random_wide"])
def test_smvm_builds_nothing_until_the_16th_product_unless_told(kind):
    """Amortisation (spmv.hpp: spmv_may_build; sparse_amd_csr_*_prepare).  A large record's faster SpMV forms
    (16-bit indices for a banded record, the two-pass copy for a wide random one) cost many products' worth of time.
    Default: the first 15 products run on what the assembly left (col32 / row_ptr: 4-byte columns, path 0) and the
    pool does not grow; the 16th builds.  prepare(A, n >= 16) builds at once, prepare(A, 1) never.  Every product
    against the oracle."""
    import re
    from sparse_amd import Context
    c = Context(0)                                                    # default options: no spmv_eager
    O, M = Oracle("f64"), mk(c, "f64")
    if kind == "banded":
        n = m = 700000
        per = 7
        i = np.arange(n, dtype=np.int64)
        r = np.repeat(i, per)
        cc = np.minimum(np.repeat(i, per) + np.tile(np.arange(per, dtype=np.int64), n), m - 1)
        built = lambda: c.get_option("spmv_last_col_bytes") == 2
    else:
        n, m = 40000, 1 << 23                                           # x = 64 MB of f64: gather-bound, flat popularity
        k = 6000000
        r = (synth.splitmix64(91, np.arange(k, dtype=np.uint64)) % np.uint64(n)).astype(np.int64)
        cc = (synth.splitmix64(92, np.arange(k, dtype=np.uint64)) % np.uint64(m)).astype(np.int64)
        built = lambda: c.get_option("spmv_last_path") in (2, 3)
    v = synth.uniform_values(93, len(r), "f64")
    Ao = O.sr_sparse(n, m, r, cc, v)
    x = synth.uniform_values(94, m, "f64")
    want = O.sr_smvm(Ao, x)
    xd = c.array(x, "f64")
    yd = c.array(np.zeros(n), "f64")

    def held():
        return int(re.search(r"device memory held: (\d+) bytes", c.report()).group(1))

    def check():
        np.testing.assert_allclose(yd.numpy(), want, rtol=1e-12, atol=0)

    # (1) no hint: light until the 16th product
    A = M.sr.sparse(n, m, (r, cc, v))
    c.sync()
    before = held()
    for i in range(15):
        M.sr.smvm_into(yd, A, xd)
        assert c.get_option("spmv_last_path") == 0 and c.get_option("spmv_last_col_bytes") == 4, i
    check()
    assert held() == before, "a product before the 16th built something"
    M.sr.smvm_into(yd, A, xd)                                         # the 16th
    assert built()
    check()
    assert held() > before
    # (2) told at once
    B = M.sr.sparse(n, m, (r, cc, v))
    M.sr.prepare(B, 1000)
    grown = held()
    M.sr.smvm_into(yd, B, xd)
    assert built()
    check()
    assert held() == grown, "the first product after prepare still built something"
    # (3) told it is a one-off: never built, however often it is multiplied after all
    C = M.sr.sparse(n, m, (r, cc, v))
    M.sr.prepare(C, 1)
    for i in range(20):
        M.sr.smvm_into(yd, C, xd)
    assert not built() and c.get_option("spmv_last_path") == 0
    check()
    with pytest.raises(Exception, match="expected_products"):
        M.sr.prepare(C, 0)
    # the csc label of the same record (vsmm, compressed.fut:223-224)
    T = M.sr.transpose(C)
    M.sc.prepare(T, 1000)
    np.testing.assert_allclose(M.sc.vsmm(x, T).numpy(), want, rtol=1e-12, atol=0)
    assert built()
    c.sync()
    c.close()


def test_smvm_is_reproducible(ctx):
    M = mk(ctx, "f64")
    r, c, v = skewed_coo(3000, 9000, "f64", 8)
    A = M.sr.sparse(3000, 9000, (r, c, v))
    x = synth.uniform_values(1, 9000, "f64")
    y0 = M.sr.smvm(A, x).numpy()
    for _ in range(3):
        assert np.array_equal(M.sr.smvm(A, x).numpy(), y0)


# ------------------------------------------------- the rest of the surface
@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_coo_dense_nnz_scale_addsub_conversion(ctx, dtype):
    O, M = Oracle(dtype), mk(ctx, dtype)
    n, m, k = 700, 500, 9000
    r, c, v = random_coo(n, m, k, dtype, seed=21, dup_frac=0.1)
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
    for got, want in zip(M.sr.coo(A), O.sr_coo(Ao)):
        assert np.array_equal(np.asarray(got), want)
    B, Bo = M.sc.sparse(n, m, (r, c, v)), O.sc_sparse(n, m, r, c, v)
    for got, want in zip(M.sc.coo(B), O.sc_coo(Bo)):
        assert np.array_equal(np.asarray(got), want)
    assert np.array_equal(M.sr.dense(A).numpy(), O.sr_dense(Ao))
    assert np.array_equal(M.sc.dense(B).numpy(), O.sc_dense(Bo))
    assert M.sr.nnz(A) == O.sr_nnz(Ao)
    s = {"f64": 2.5, "f32": 2.5, "i32": -3}[dtype]
    assert_same_record(M.sr.scale(s, A), O.sr_scale(s, Ao))
    r2, c2, v2 = random_coo(n, m, k // 3, dtype, seed=22)
    A2, A2o = M.sr.sparse(n, m, (r2, c2, v2)), O.sr_sparse(n, m, r2, c2, v2)
    assert_same_record(M.sr.add(A, A2), O.sr_add(Ao, A2o))
    assert_same_record(M.sr.sub(A, A2), O.sr_sub(Ao, A2o))
    d = M.sr.sub(A, A)                       # explicit zeros are kept, nnz counts none
    assert d.stored_dims()[2] == Ao["nnz"] and M.sr.nnz(d) == 0
    # physical conversion and back
    T = M.sr.to_sc(A)
    assert_same_record(T, O.sr_to_sc(Ao))
    assert_same_record(M.sc.to_sr(T), Ao)
    # zero-cost transpose: same record, other label
    assert_same_record(M.sr.transpose(A), Ao)
    assert np.array_equal(M.sc.dense(M.sr.transpose(A)).numpy(), O.sr_dense(Ao).T)


@pytest.mark.parametrize("tiles,big", [(1, False), (3, False), (5000, True)])
def test_coo_conversion_addsub_when_entries_fill_whole_tiles_and_empty_rows_trail(ctx, tiles, big):
    """A record whose entry count is a multiple of the tile size (1024) and whose last rows are empty: those rows
    start at offset 1024 of the last tile, one past the row marks a tile's workgroup keeps in LDS (tile_rows:
    sr.coo, the key packing of the conversions -- with and without digit counts, i.e. small and large records --
    and of + / -).  Every operation against the oracle."""
    dtype = "f64"
    O, M = Oracle(dtype), mk(ctx, dtype)
    nnz = tiles * 1024
    per = 8
    n_full = nnz // per
    n, m = n_full + 300, n_full + per
    i = np.arange(n_full, dtype=np.int64)
    r = np.repeat(i, per)
    c = np.repeat(i, per) + np.tile(np.arange(per, dtype=np.int64), n_full)
    v = synth.uniform_values(61, nnz, dtype)
    A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
    assert A.stored_dims()[2] == nnz and Ao["row_off"][-1] == nnz and Ao["row_off"][n_full - 1] == nnz
    for got, want in zip(M.sr.coo(A), O.sr_coo(Ao)):
        assert np.array_equal(np.asarray(got), want)
    T = M.sr.to_sc(A)
    assert_same_record(T, O.sr_to_sc(Ao))
    assert_same_record(M.sc.to_sr(T), Ao)
    if not big:
        assert_same_record(M.sr.add(A, A), O.sr_add(Ao, Ao))
        assert_same_record(M.sr.sub(A, M.sr.eye(n, m)), O.sr_sub(Ao, O.sr_eye(n, m)))
    else:
        S = M.sr.add(A, A)
        ro, ci, va = (np.asarray(f) for f in S.fields())
        assert np.array_equal(ro, Ao["row_off"]) and np.array_equal(ci, Ao["col_idx"]) and np.array_equal(va, 2 * Ao["vals"])


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("shape", [(3000, 2000, 70000, 50000), (5, 100000, 60000, 1), (100000, 100000, 40000, 90000),
                                   (64, 64, 4096, 0), (1 << 17, 1 << 33, 50000, 50000)],
                         ids=["overlapping", "one_entry_second", "second_longer", "empty_second", "wide_keys"])
def test_add_sub_merge_the_two_records_against_the_oracle(ctx, dtype, shape):
    """sr.+ / sr.- = sparse (coo a ++ coo (scale -1 b)) (compressed.fut:179-183).  Both halves are in key order, so the
    library merges them (merge path) instead of sorting: same record as the oracle, and as its own sort (add_sort)."""
    n, m, ka, kb = shape
    O, M = Oracle(dtype), mk(ctx, dtype)
    ra = (synth.splitmix64(1, np.arange(ka, dtype=np.uint64)) % np.uint64(n)).astype(np.int64)
    ca = (synth.splitmix64(2, np.arange(ka, dtype=np.uint64)) % np.uint64(m)).astype(np.int64)
    va = synth.uniform_values(3, ka, dtype)
    # the second record shares about half of its coordinates with the first
    kb_sh = min(kb // 2, ka)
    rb = np.concatenate([ra[:kb_sh], (synth.splitmix64(4, np.arange(kb - kb_sh, dtype=np.uint64)) % np.uint64(n)).astype(np.int64)])
    cb = np.concatenate([ca[:kb_sh], (synth.splitmix64(5, np.arange(kb - kb_sh, dtype=np.uint64)) % np.uint64(m)).astype(np.int64)])
    vb = synth.uniform_values(6, kb, dtype)
    A, Ao = M.sr.sparse(n, m, (ra, ca, va)), O.sr_sparse(n, m, ra, ca, va)
    B, Bo = M.sr.sparse(n, m, (rb, cb, vb)), O.sr_sparse(n, m, rb, cb, vb)
    for op, oop in ((M.sr.add, O.sr_add), (M.sr.sub, O.sr_sub)):
        want = oop(Ao, Bo)
        assert_same_record(op(A, B), want)
        assert_same_record(op(B, A), oop(Bo, Ao))
        ctx.set_option("add_sort", 1)
        try:
            assert_same_record(op(A, B), want)
        finally:
            ctx.set_option("add_sort", 0)


@pytest.mark.parametrize("dtype", ["f64", "i32"])
def test_constructors(ctx, dtype):
    O, M = Oracle(dtype), mk(ctx, dtype)
    for n, m in ((3, 2), (2, 3), (5, 5), (0, 0), (0, 1), (1, 0), (300, 7)):
        assert_same_record(M.sr.eye(n, m), O.sr_eye(n, m))
        assert_same_record(M.sc.eye(n, m), O.sc_eye(n, m))
        assert_same_record(M.sr.zero(n, m), O.sr_zero(n, m))
    v = synth.uniform_values(2, 1000, dtype)
    assert_same_record(M.sr.diag(v), O.sr_diag(v))


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
@pytest.mark.parametrize("k", [1, 8, 64, 100])
@pytest.mark.parametrize("algo", [0, 2])
def test_smm_matches_oracle(ctx, dtype, k, algo):
    O, M = Oracle(dtype), mk(ctx, dtype)
    ctx.set_option("spmm_algo", algo)
    try:
        for n, m, gen in ((900, 700, lambda: skewed_coo(900, 700, dtype, 31)),
                          (3000, 3000, lambda: synth.banded_coo(3000, 13, dtype)),
                          (5, 4000, lambda: random_coo(5, 4000, 9000, dtype, seed=33))):
            r, c, v = gen()
            A, Ao = M.sr.sparse(n, m, (r, c, v)), O.sr_sparse(n, m, r, c, v)
            B = synth.uniform_values(32, m * k, dtype).reshape(m, k)
            close(M.sr.smm(A, B), O.sr_smm(Ao, B), dtype)
        Z = M.sr.zero(7, 5)
        assert np.array_equal(M.sr.smm(Z, np.ones((5, k), NP[dtype])).numpy(), np.zeros((7, k), NP[dtype]))
    finally:
        ctx.set_option("spmm_algo", 0)


def _banded_with_holes(n, hb, dtype, seed):
    """Band of half-width hb with some rows emptied, some thinned and one row much longer than a tile."""
    r, c, v = synth.banded_coo(n, hb, dtype)
    keep = np.ones(len(r), bool)
    keep[(r % 97 == 5) | ((r >= 700) & (r < 760))] = False           # empty rows, and a run of them
    keep[(r % 11 == 3) & (c % 2 == 0)] = False                         # thinned rows
    extra_c = np.arange(0, min(n, 2600), dtype=np.int64)               # one row spanning several tiles
    extra_r = np.full(len(extra_c), 1234 if n > 1234 else 0, np.int64)
    r = np.concatenate([r[keep], extra_r])
    c = np.concatenate([c[keep], extra_c])
    v = synth.uniform_values(seed, len(r), dtype)
    return r, c, v


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("k", [64, 16, 40, 130, 8, 48, 3])
def test_smm_matrix_core_tiles(ctx, dtype, k):
    """Banded tiles with long rows take the dense-block (MFMA) path of spmm_stream; the same record with
    spmm_algo 3 never does.  Both against the oracle, on a band with empty rows, thinned rows, rows crossing
    tile ends and a row longer than a tile (whose tiles cannot use the dense block); panels narrower than one
    wavefront's 16 columns and panels that are not a multiple of 16 among them."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    for n, hb in ((5000, 13), (3000, 30), (777, 9)):
        r, c, v = _banded_with_holes(n, hb, dtype, seed=70 + hb)
        A, Ao = M.sr.sparse(n, n, (r, c, v)), O.sr_sparse(n, n, r, c, v)
        B = synth.uniform_values(71, n * k, dtype).reshape(n, k)
        want = O.sr_smm(Ao, B)
        got = M.sr.smm(A, B).numpy()
        close(got, want, dtype)
        assert got.tobytes() == M.sr.smm(A, B).numpy().tobytes()      # reproducible
        ctx.set_option("spmm_algo", 3)
        try:
            close(M.sr.smm(A, B), want, dtype)
        finally:
            ctx.set_option("spmm_algo", 0)


@pytest.mark.parametrize("k", [64, 24])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_smm_matrix_core_tiles_keep_nonfinite_rows_of_B_local(ctx, dtype, k):
    """The dense block is zero-filled; 0 * inf must not leak into rows that do not reference the row of B
    holding the inf / nan: same pattern of non-finite results as the oracle, finite values within tolerance."""
    O, M = Oracle(dtype), mk(ctx, dtype)
    n, hb = 4000, 13
    r, c, v = synth.banded_coo(n, hb, dtype)
    keep = ~((c == 1000) & (r != 1003)) & ~((c == 2500) & (r % 2 == 0))   # column 1000 referenced by one row only
    r, c, v = r[keep], c[keep], v[keep]
    A, Ao = M.sr.sparse(n, n, (r, c, v)), O.sr_sparse(n, n, r, c, v)
    B = synth.uniform_values(72, n * k, dtype).reshape(n, k).copy()
    B[1000, :] = np.inf
    B[2500, 7] = np.nan
    B[2500, 40 if k > 40 else 19] = -np.inf
    with np.errstate(invalid="ignore"):
        want = O.sr_smm(Ao, B)
    got = M.sr.smm(A, B).numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    assert fin.sum() > 0.9 * fin.size and (~fin).sum() > 0
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL[dtype], atol=0)
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])


@pytest.mark.parametrize("dtype", ["f64", "i32"])
def test_smsmm_matches_oracle(ctx, dtype):
    Oc, M = OracleCompressed(dtype), mk(ctx, dtype)
    n, m, k = 40, 30, 25
    r, c, v = random_coo(n, m, 300, dtype, seed=41)
    r2, c2, v2 = random_coo(m, k, 200, dtype, seed=42)
    A, B = M.sr.sparse(n, m, (r, c, v)), M.sc.sparse(m, k, (r2, c2, v2))
    Ao, Bo = Oc.sr.sparse(n, m, (r, c, v)), Oc.sc.sparse(m, k, (r2, c2, v2))
    want = Oc.smsmm(Ao, Bo)
    assert_same_record(M.smsmm(A, B), want, exact_vals=(dtype == "i32"), dtype=dtype)


def test_from_fields_roundtrip(ctx):
    from sparse_amd.compressed import SparseError
    O, M = Oracle("f64"), mk(ctx, "f64")
    r, c, v = random_coo(300, 400, 5000, "f64", seed=51)
    Ao = O.sr_sparse(300, 400, r, c, v)
    A = M.sr.from_fields(300, 400, Ao["row_off"], Ao["col_idx"], Ao["vals"])
    assert_same_record(A, Ao)
    x = synth.uniform_values(1, 400, "f64")
    close(M.sr.smvm(A, x), O.sr_smvm(Ao, x), "f64")
    bad = Ao["col_idx"].copy()
    bad[3] = 400
    with pytest.raises(SparseError):
        M.sr.from_fields(300, 400, Ao["row_off"], bad, Ao["vals"])
    bad_ro = Ao["row_off"].copy()
    bad_ro[10] = bad_ro[9] - 1 if bad_ro[9] > 0 else 10**9
    with pytest.raises(SparseError):
        M.sr.from_fields(300, 400, bad_ro, Ao["col_idx"], Ao["vals"])


@pytest.mark.parametrize("dtype", ["f64", "f32", "i32"])
def test_store_restore_opaque_roundtrip(ctx, dtype):
    """futhark_store_opaque_* -> futhark_restore_opaque_* through the raw C ABI: the restored handle's
    fields are bit-equal (also for a csc handle, an empty record and the size-query / caller-buffer forms),
    and malformed bytes are refused."""
    from sparse_amd.compressed import SparseError
    M, O = mk(ctx, dtype), Oracle(dtype)
    n, m = 700, 900
    r, c, v = random_coo(n, m, 9000, dtype, seed=77)
    A = M.sr.sparse(n, m, (r, c, v))
    blob = A.store()
    A2 = M.sr.restore(blob)
    for a, b in zip(fields(A), fields(A2)):
        assert np.array_equal(a, b)
    assert_same_record(A2, O.sr_sparse(n, m, r, c, v))
    x = synth.uniform_values(5, m, dtype)
    assert np.array_equal(M.sr.smvm(A2, x).numpy(), M.sr.smvm(A, x).numpy())
    # csc handle: the stored record is that of the transpose
    B = M.sc.sparse(n, m, (r, c, v))
    B2 = M.sc.restore(B.store())
    assert B2.stored_dims() == B.stored_dims() == (m, n, B.stored_dims()[2])
    for a, b in zip(fields(B), fields(B2)):
        assert np.array_equal(a, b)
    # empty record
    Z2 = M.sr.restore(M.sr.zero(3, 4).store())
    assert Z2.stored_dims() == (3, 4, 0)
    # size query (p == NULL) and a caller-provided buffer
    L = ctx.L
    f = getattr(L, "futhark_store_opaque_csr_" + dtype)
    size = ctypes.c_size_t()
    assert f(ctx.h, A.h, None, ctypes.byref(size)) == 0 and size.value == len(blob)
    buf = ctypes.create_string_buffer(size.value)
    pbuf = ctypes.c_void_p(ctypes.addressof(buf))
    assert f(ctx.h, A.h, ctypes.byref(pbuf), ctypes.byref(size)) == 0
    assert pbuf.value == ctypes.addressof(buf) and buf.raw == blob
    # refused: wrong magic, wrong element type, a record that is not normalised
    with pytest.raises(SparseError):
        M.sr.restore(b"X" + blob[1:])
    other = mk(ctx, "f32" if dtype != "f32" else "f64")
    with pytest.raises(SparseError):
        other.sr.restore(blob)
    bad = bytearray(blob)
    hdr = 8 + 8 + 24
    bad[hdr:hdr + 8] = np.int64(10**6).tobytes()      # row_off[0] beyond nnz
    with pytest.raises(SparseError):
        M.sr.restore(bytes(bad))


def test_wide_col_option_keeps_parity(ctx):
    """col_idx kept as i64 on device (the layout used when m > 2^31-1) gives the same results."""
    O, M = Oracle("f64"), mk(ctx, "f64")
    ctx.set_option("wide_col", 1)
    try:
        r, c, v = random_coo(2000, 3000, 30000, "f64", seed=61)
        A, Ao = M.sr.sparse(2000, 3000, (r, c, v)), O.sr_sparse(2000, 3000, r, c, v)
        assert_same_record(A, Ao)
        x = synth.uniform_values(1, 3000, "f64")
        close(M.sr.smvm(A, x), O.sr_smvm(Ao, x), "f64")
        assert_same_record(M.sr.to_sc(A), O.sr_to_sc(Ao))
        B = synth.uniform_values(2, 3000 * 4, "f64").reshape(3000, 4)
        close(M.sr.smm(A, B), O.sr_smm(Ao, B), "f64")
    finally:
        ctx.set_option("wide_col", 0)


def test_c_harness_written_against_generated_api(ctx):
    """integration/harness_example.c uses only the generated-library convention."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "harness_example")
    assert os.path.exists(exe), "build first (__graft_entry__.build)"
    out = subprocess.run([exe], capture_output=True, timeout=100)
    assert out.returncode == 0, out.stderr.decode()
    assert b"harness ok" in out.stdout


def test_matrix_market_file_to_spmv(ctx, tmp_path):
    """File -> triples (C++ loader) -> sparse -> smvm, against the oracle fed from the same file."""
    from sparse_amd.io import read_mtx, write_mtx
    O, M = Oracle("f64"), mk(ctx, "f64")
    n, m = 3000, 2500
    r, c, v = random_coo(n, m, 40000, "f64", seed=12)
    p = tmp_path / "a.mtx"
    write_mtx(p, n, m, (r, c, v))
    nn, mm, coo = read_mtx(p, threads=4)
    assert (nn, mm) == (n, m)
    A = M.sr.sparse(nn, mm, coo)
    assert_same_record(A, O.sr_sparse(n, m, r, c, v))
    x = synth.uniform_values(2, m, "f64")
    close(M.sr.smvm(A, x), O.sr_smvm(O.sr_sparse(n, m, r, c, v), x), "f64")

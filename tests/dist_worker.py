"""Worker for tests/test_distributed_cpu.py: one rank of a gloo world on the CPU.

Exercises sparse_amd.distributed's partitioning and exchange plan (pure torch +
torch.distributed, the same code the RCCL path runs); the local product is done
by the CPU oracle since there is no GPU here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle  # noqa: E402
from sparse_amd import synth  # noqa: E402
from sparse_amd.distributed import (ExchangePlan, coarse_row_splitters, partition_rows_equal_nnz,  # noqa: E402
                                    redistribute_coo)


def run_case(name, n, r, c, v, force_mode, bounds=None):
    rank, world = dist.get_rank(), dist.get_world_size()
    O = Oracle("f64")
    G = O.sr_sparse(n, n, r, c, v)
    x = 1.0 + (np.arange(n) % 7) / 8.0
    y_ref = O.sr_smvm(G, x)
    if bounds is None:
        bounds = partition_rows_equal_nnz(G["row_off"], world)
    lo, hi = bounds[rank], bounds[rank + 1]
    gr, gc, gv = O.sr_coo(G)
    sel = (gr >= lo) & (gr < hi)
    lr, lc, lv = gr[sel], gc[sel], gv[sel]
    plan = ExchangePlan(dist, bounds, torch.unique(torch.from_numpy(lc)), "cpu", force_mode)
    lc2 = plan.remap_cols(torch.from_numpy(lc)).numpy()
    L = O.sr_sparse(hi - lo, plan.x_len, lr - lo, lc2, lv)
    x_ext, x_own = plan.new_x(torch.float64)
    x_own.copy_(torch.from_numpy(x[lo:hi]))
    plan.exchange(x_ext)
    y = O.sr_smvm(L, x_ext.numpy())
    assert np.array_equal(y, y_ref[lo:hi]), (name, rank, plan.mode)
    if plan.mode == "allgather":                       # the all-pairs form of the same exchange
        plan.allgather_impl = "p2p"
        x_ext[:lo].zero_()
        x_ext[hi:].zero_()
        plan.exchange(x_ext)
        assert np.array_equal(O.sr_smvm(L, x_ext.numpy()), y_ref[lo:hi]), (name, rank, "p2p all-gather")
        plan.allgather_impl = "collective"
    # a second exchange after x changed must refresh the remote entries
    x2 = 2.0 * x
    x_own.copy_(torch.from_numpy(x2[lo:hi]))
    plan.exchange(x_ext)
    assert np.array_equal(O.sr_smvm(L, x_ext.numpy()), 2.0 * y_ref[lo:hi]), (name, rank, "refresh")
    return plan.mode, int(plan.x_len)


def run_assembly_case(name, n, m, r, c, v, bins_log2=16):
    """Distributed assembly: each rank starts from an arbitrary (strided) chunk of the global
    triple list; after the exchange + local assembly the pieces must be, bit for bit, the row
    blocks of the single-process result (duplicates summed in global input order)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    O = Oracle("f64")
    G = O.sr_sparse(n, m, r, c, v)
    # contiguous chunks of unequal length, so that the global input order is rank-major
    cuts = [0] + [(len(r) * (k + 1) * (k + 2)) // (world * (world + 1)) for k in range(world)]
    sl = slice(cuts[rank], cuts[rank + 1])
    tr, tc, tv = (torch.from_numpy(np.ascontiguousarray(a[sl])) for a in (r, c, v))
    bounds = coarse_row_splitters(dist, n, tr, world, bins_log2)
    assert bounds[0] == 0 and bounds[-1] == n and all(a <= b for a, b in zip(bounds, bounds[1:])), bounds
    lr, lc, lv = redistribute_coo(dist, bounds, tr, tc, tv)
    lo, hi = bounds[rank], bounds[rank + 1]
    assert lr.numel() == 0 or (int(lr.min()) >= lo and int(lr.max()) < hi)
    L = O.sr_sparse(hi - lo, m, lr.numpy() - lo, lc.numpy(), lv.numpy())
    # global row_off = local inclusive scan + entries on lower ranks
    mine = torch.tensor([L["nnz"]], dtype=torch.int64)
    allv = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allv, mine)
    off = sum(int(x.item()) for x in allv[:rank])
    assert sum(int(x.item()) for x in allv) == G["nnz"], name
    assert np.array_equal(L["row_off"] + off, G["row_off"][lo:hi]), (name, rank)
    assert np.array_equal(L["col_idx"], G["col_idx"][off:off + L["nnz"]]), (name, rank)
    assert L["vals"].tobytes() == G["vals"][off:off + L["nnz"]].tobytes(), (name, rank)
    # balance: no part holds more than its share plus one histogram bin's worth of triples
    tot = torch.tensor([lr.numel()], dtype=torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.MAX)
    return int(tot.item())


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = 24
    n = g * g * world
    r, c, v = synth.laplacian_2d_coo(g)            # square grid first
    mode, xlen = run_case("lap-square", g * g, r, c, v, None)
    assert mode == ("halo" if world > 1 else "none")
    # rectangular stencil: world*g x g grid, rows split evenly -> halo of g entries per neighbour
    i = np.arange(n, dtype=np.int64)
    gx, gy = i % g, i // g
    rows = [i, i[gx > 0], i[gx < g - 1], i[gy > 0], i[gy < world * g - 1]]
    cols = [i, i[gx > 0] - 1, i[gx < g - 1] + 1, i[gy > 0] - g, i[gy < world * g - 1] + g]
    vals = [np.full(len(rows[0]), 4.0)] + [np.full(len(q), -1.0) for q in rows[1:]]
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    even = [k * g * g for k in range(world + 1)]
    mode, xlen = run_case("lap-rect", n, r, c, v, None, even)
    if world > 1:
        assert mode == "halo"
        nb = (1 if rank > 0 else 0) + (1 if rank < world - 1 else 0)
        assert xlen == g * g + nb * g, (xlen, g)
    # random matrix: every rank references most of x -> all-gather (equal slices: one collective)
    rr, cc, _ = synth.random_uniform_coo(n, n, 20 * n, "f64", seed=3)
    vv = (synth.splitmix64(9, np.arange(len(rr), dtype=np.uint64)) % np.uint64(5)).astype(np.float64) - 2.0
    mode, _ = run_case("random-even", n, rr, cc, vv, None, even)
    assert mode == ("allgather" if world > 1 else "none")
    # unequal slices (equal-nnz splitters of a skewed matrix) -> point-to-point all-gather
    rr2 = (rr.astype(np.float64) ** 2 / n).astype(np.int64)
    mode, _ = run_case("random-skewed", n, rr2, cc, vv, None)
    # forcing either path gives the same answer
    run_case("rect-forced-allgather", n, r, c, v, "allgather", even)
    run_case("random-forced-halo", n, rr, cc, vv, "halo", even)
    # ---- distributed assembly
    k = 40 * n
    idx = np.arange(k, dtype=np.uint64)
    ar = (synth.splitmix64(11, idx) % np.uint64(n)).astype(np.int64)
    ac = (synth.splitmix64(12, idx) % np.uint64(n // 4)).astype(np.int64)      # many duplicates
    av = 0.5 + (synth.splitmix64(13, idx) >> np.uint64(11)).astype(np.float64) / 2.0 ** 53
    worst = run_assembly_case("asm-random-dups", n, n // 4, ar, ac, av)
    assert worst <= k // world + k // 8, (worst, k)
    run_assembly_case("asm-coarse-bins", n, n // 4, ar, ac, av, bins_log2=3)
    skew = (ar.astype(np.float64) ** 3 / float(n) ** 2).astype(np.int64)       # most triples in the first rows
    run_assembly_case("asm-skewed", n, n // 4, skew, ac, av)
    run_assembly_case("asm-empty", n, n, ar[:0], ac[:0], av[:0])
    run_assembly_case("asm-one-row", n, n // 4, np.full(k, n - 1, dtype=np.int64), ac, av)
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

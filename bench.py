#!/usr/bin/env python3
"""bench.py -- CSR f64 SpMV on N MI355X (BASELINE.json's metric).

Headline leg (`value`, `ms_per_step`, `roofline`): the 5-point Laplacian of a 4096^2 grid per GPU
(BASELINE configs[1]).  One "step" = one y = A x over the whole matrix (all ranks).  The matrix is built
through the product path itself (shuffled COO -> sr.sparse on the device); x and the matrix are resident in
HBM before the timed region; a step allocates nothing (sparse_amd_csr_f64_smvm_into on handles made once).
N > 1 (one process per GPU, torch.distributed over RCCL): weak scaling -- rank k owns the k-th block of
4096*4096 rows of the (N*4096) x 4096 grid's matrix and the matching slice of x; each step exchanges the x
entries other ranks need (sparse_amd.distributed: halo transfers overlapped with the owned-column block),
then runs the same SpMV kernel on its rows.

Second leg (`c5` object of the same JSON line; --no-c5 skips it): what north_star scales on -- BASELINE
configs[4], COO -> CSR assembly + SpMV of a 10^9-triple uniform random 1e7 x 1e7 f64 matrix, STRONG scaling:
the same global matrix at every N.  Each rank generates 1/N of the triple list from the counter PRNG
(synth.random_uniform_coo_counter_device), the ranks exchange triples so that each owns a row block
(DistCsr.from_coo_chunks: one all-to-all-v, then the single-GPU assembly), every step all-gathers x over
RCCL and multiplies.  Parity gate on integer-valued data (sums exact in any order): y equals torch's own
scatter-add of the generated triples, bit for bit, on every rank.  Assembly, the exchange alone and the
kernel alone are timed separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GRID = 4096
C5_N = 10**7
C5_TRIPLES = 10**9


def laplacian_csr_host(g):
    """The stencil's record built directly (rows are regular: no sort): row_off inclusive, col_idx, vals."""
    import numpy as np
    n = g * g
    i = np.arange(n, dtype=np.int64)
    gx, gy = i % g, i // g
    offs = np.array([-g, -1, 0, 1, g], dtype=np.int64)
    ok = np.stack([gy > 0, gx > 0, np.ones(n, bool), gx < g - 1, gy < g - 1], axis=1)     # [n][5], ascending columns
    cols = (i[:, None] + offs[None, :])[ok]
    vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (n, 5))[ok]
    row_off = np.cumsum(ok.sum(axis=1)).astype(np.int64)
    return {"n": n, "m": n, "nnz": int(row_off[-1]), "row_off": row_off, "col_idx": np.ascontiguousarray(cols),
            "vals": np.ascontiguousarray(vals)}


def cpu_baseline(g, seconds=12.0):
    """The oracle's SpMV (a port of compressed.fut:134-146, rows over OpenMP threads) timed on this host on
    the bench's own matrix (the 4096^2 stencil), a bounded number of products."""
    import numpy as np
    from oracle.oracle import Oracle, lib
    A = laplacian_csr_host(g)
    n = A["n"]
    x = 1.0 + (np.arange(n) % 7) / 8.0
    threads = min(16, os.cpu_count() or 1) if lib().oracle_has_openmp() else 1
    O = Oracle("f64")
    O.sr_smvm(A, x, threads=threads)      # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        O.sr_smvm(A, x, threads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds or reps >= 4000:
            break
    return {"value": round(2.0 * A["nnz"] * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": threads,
            "kind": "port",
            "sample": "the bench's own matrix: 5-point Laplacian %d^2 grid (n=%d, nnz=%d), %d SpMVs in %.1f s, "
                      "OpenMP row loop of the oracle" % (g, n, A["nnz"], reps, el)}


def timed_steps(torch, dist, fn, steps):
    """barrier + synchronize on both sides; returns (wall seconds, device ms per step)."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps


def max_over_ranks(torch, dist, dev, v):
    if dist is None:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


SETTLE = 300   # products of the reproducibility gate (about 60 ms on the C2 workload)


def c5_leg(args, torch, dist, world, rank, dev, ctx, M):
    """BASELINE configs[4]: assembly + SpMV of the 1e9-triple random matrix, row-sharded, strong scaling."""
    from sparse_amd import synth
    from sparse_amd import distributed as sd
    n, total = args.c5_n, args.c5_triples
    lo, hi = total * rank // world, total * (rank + 1) // world
    r, c, v = synth.random_uniform_coo_counter_device(n, n, lo, hi - lo, dev, integer_values=True)
    # once untimed: the first assembly of this size pays for the device allocations behind the library's pool and
    # torch's (hipMalloc of tens of GB) and for RCCL's connections
    D = sd.DistCsr.from_coo_chunks(M, dist, n, n, r, c, v)
    del D
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    D = sd.DistCsr.from_coo_chunks(M, dist, n, n, r, c, v)
    ctx.sync()
    torch.cuda.synchronize()
    asm_s = max_over_ranks(torch, dist, dev, time.perf_counter() - t0)
    rlo, rhi = D.bounds[rank], D.bounds[rank + 1]
    D.x_own.copy_(1.0 + (torch.arange(rlo, rhi, device=dev) % 7).double())          # integer-valued x
    # ---- parity gate: torch's own scatter-add of the generated triples (global, summed over the ranks)
    y = D.smvm().clone()
    path = ctx.get_option("spmv_last_path")
    xg = 1.0 + (torch.arange(n, device=dev) % 7).double()
    want = torch.zeros(n, dtype=torch.float64, device=dev)
    chunk = 1 << 28
    for s in range(0, r.numel(), chunk):        # sums of small integers: exact in any order
        want.index_add_(0, r[s:s + chunk], v[s:s + chunk] * xg[c[s:s + chunk]])
    if dist is not None:
        dist.all_reduce(want)
    assert torch.equal(y, want[rlo:rhi]), "C5 leg: SpMV differs from the scatter-add of the generated triples"
    del want, xg, r, c, v, y
    torch.cuda.empty_cache()
    nnz_glob = D.nnz
    if dist is not None:
        t = torch.tensor([D.nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        nnz_glob = int(t.item())
    steps = max(3, args.steps // 10)
    tuned = D.tune_exchange()                 # one collective or all-pairs transfers: both timed, the faster kept
    for _ in range(2):
        D.smvm()
    el, _ = timed_steps(torch, dist, D.smvm, steps)
    el = max_over_ranks(torch, dist, dev, el)
    D.exchange()
    _, kern_ms = timed_steps(torch, None, D.local_smvm, steps)
    ex_ms = 0.0
    if dist is not None:
        ex_el, _ = timed_steps(torch, dist, D.exchange, steps)
        ex_ms = ex_el / steps * 1e3
    ms = el / steps * 1e3
    return {"workload": "COO->CSR assembly + SpMV, uniform random %d x %d f64, %d triples, row-sharded over %d GPU(s) "
                        "(BASELINE configs[4]); strong scaling: the same matrix at every N" % (n, n, total, world),
            "n_gpus": world, "nnz_total": nnz_glob, "steps": steps, "ms_per_step": round(ms, 4),
            "GFLOPs": round(2.0 * nnz_glob / ms / 1e6, 1),
            "assembly_ms": round(asm_s * 1e3, 2), "assembly_Mtriples_per_s": round(total / asm_s / 1e6, 1),
            "x_exchange": D.mode, "x_exchange_impl": getattr(D.plan, "allgather_impl", None),
            "x_exchange_tuning_ms": {k: round(v * 1e3, 4) for k, v in tuned.items()},
            "x_exchange_ms_alone": round(ex_ms, 4), "kernel_ms_alone": round(kern_ms, 4),
            "spmv_path": {0: "stream", 1: "hot-column renumbering", 2: "column blocks", 3: "two passes"}.get(path, str(path)),
            "rows_this_rank": rhi - rlo, "parity": "y == scatter-add of the generated triples (integer-valued data), bit for bit"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the 1e9-triple strong-scaling leg")
    ap.add_argument("--algo", type=int, default=0)
    # rehearsal switches (defaults are the measured configuration)
    ap.add_argument("--grid", type=int, default=GRID, help="grid edge per GPU (default 4096 = BASELINE configs[1])")
    ap.add_argument("--c5-n", type=int, default=C5_N)
    ap.add_argument("--c5-triples", type=int, default=C5_TRIPLES)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo rehearses N>1 on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--exchange", default=None, choices=["halo", "allgather"], help="force the x exchange mode")
    ap.add_argument("--watchdog", type=int, default=0, help="dump all stacks and exit after this many seconds")
    args = ap.parse_args()
    if args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.same_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if dist is not None and args.backend == "gloo":
        from sparse_amd.distributed import HostStagedDist
        dist = HostStagedDist(dist)      # rehearsal with ranks sharing a GPU: collectives go through CPU copies
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from sparse_amd import Context, mk_compressed, synth
    from sparse_amd import distributed as sd
    ctx = Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.algo:
        ctx.set_option("spmv_algo", args.algo)
    M = mk_compressed("f64", ctx)

    # ---- build this rank's rows through the product path
    g = args.grid
    n_loc = g * g
    n_glob = n_loc * world
    if world == 1:
        r, c, v = synth.laplacian_2d_coo_device(g, dev)
        torch.cuda.synchronize()
        t_asm0 = time.perf_counter()
        A = M.sr.sparse(n_loc, n_loc, (r, c, v))
        ctx.sync()
        asm_s = time.perf_counter() - t_asm0
        nnz0 = r.numel()
        del r, c, v
        D = None
        x = synth.laplacian_x_device(0, n_loc, dev)
        y = torch.empty(n_loc, dtype=torch.float64, device=dev)
        xh, yh = ctx.array(x, "f64"), ctx.array(y, "f64")       # handles made once: a step is one C call
    else:
        r, c, v = synth.laplacian_2d_coo_device(g, dev, row_lo=rank * n_loc, row_hi=(rank + 1) * n_loc, gy=g * world)
        nnz0 = r.numel()
        torch.cuda.synchronize()
        t_asm0 = time.perf_counter()
        D = sd.DistCsr.from_local_coo(M, dist, n_glob, n_glob, rank * n_loc, (rank + 1) * n_loc, r, c, v,
                                       force_mode=args.exchange)
        ctx.sync()
        asm_s = time.perf_counter() - t_asm0
        del r, c, v
        A = D.A
        D.x_own.copy_(synth.laplacian_x_device(rank * n_loc, (rank + 1) * n_loc, dev))
        x = D.x_own
    torch.cuda.empty_cache()
    nnz = A.stored_dims()[2] if D is None else D.nnz

    def step():
        if D is None:
            M.sr.smvm_into(yh, A, xh)
            return y
        return D.smvm()

    # ---- parity gate on the exact workload (size-independent property: the stencil itself)
    got = step()
    if world == 1:
        want = synth.laplacian_2d_apply_device(g, x)
    else:
        want = synth.laplacian_2d_apply_rows(g, g * world, rank * n_loc, (rank + 1) * n_loc, dev)
    assert torch.equal(got, want), "SpMV result differs from the stencil applied directly"
    del want
    # ---- reproducibility gate: the product is claimed bit-reproducible (no atomics on this path); SETTLE further
    # products must give the same bits.  Untimed, before the W warm-up steps, and reported as "settle_steps": on a
    # fresh card it is also what brings the clocks to their steady state (a 5 + 20 step run straight after the
    # assembly measured 0.206 ms per step where 200 steps measure 0.194).
    ref_y = got.clone()
    for _ in range(SETTLE):
        step()
    assert torch.equal(step(), ref_y), "SpMV is not reproducible bit for bit"
    del ref_y

    for _ in range(args.warmup):
        step()
    el, dev_ms = timed_steps(torch, dist, step, args.steps)

    # kernel-only timing (no exchange) for the roofline entry
    if D is not None:
        xf = D.exchange()
        _, kern_ms = timed_steps(torch, None, lambda: D.local_smvm(xf), args.steps)
        ib = ctx.get_option("spmv_last_col_bytes")     # of the dominant kernel (the last product was local_smvm)
        m_loc = D.x_len
        ex_el, _ = timed_steps(torch, dist, D.exchange, args.steps)     # wall clock: transfers run on the communicator's stream
        ex_ms = ex_el / args.steps * 1e3
    else:
        kern_ms = dev_ms
        ib = ctx.get_option("spmv_last_col_bytes")
        m_loc = n_loc
        ex_ms = 0.0

    el = max_over_ranks(torch, dist, dev, el)
    if dist is not None:
        tot = torch.tensor([nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        nnz_glob = int(tot.item())
    else:
        nnz_glob = nnz

    # ---- second leg: the 1e9-triple matrix north_star scales on (all ranks take part)
    lap_mode = "none" if D is None else D.mode
    c5 = None
    if not args.no_c5:
        del A, D, got, x
        if world == 1:
            del xh, yh, y
        else:
            del xf
        ctx.clear_caches()
        torch.cuda.empty_cache()
        c5 = c5_leg(args, torch, dist, world, rank, dev, ctx, M)

    if rank == 0:
        ms = el / args.steps * 1e3
        gflops = 2.0 * nnz_glob / (el / args.steps) / 1e9
        # algorithmic bytes of one launch on this rank, as resident on the device and read by the kernel:
        # entries (8 B value + ib B column index; ib = 2: 16-bit tile-relative columns + 4 B per 1024-entry tile)
        # + row offsets rb B (2: tile-relative u16 row_rel) and y 8 B per row + x 8 B per referenced column
        ntl = (nnz + 1023) // 1024
        rb = 2 if ntl > 4096 else 8                    # small records (launch-bound path) read row_ptr itself
        alg_bytes = nnz * (8 + ib) + (ntl * 4 if ib == 2 else 0) + n_loc * rb + n_loc * 8 + m_loc * 8
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and g == GRID:   # the PMC pass was taken on the 4096^2 per-GPU workload
            try:
                traffic = json.load(open(tpath)).get("spmv_c2_f64_hbm_bytes_per_launch")
                traffic_src = "stored: profiles/pmc_traffic.json (rocprofv3 --pmc passes of bench/pmc_traffic.sh), not measured in this run"
            except Exception:
                traffic = None
        out = {
            "metric": "CSR f64 SpMV GFLOP/s + achieved HBM GB/s (% peak) at 1/2/4/8 MI355X",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D 5-point Laplacian CSR f64, %d^2 grid per GPU, SpMV (BASELINE configs[1])" % g,
                       "rows_per_gpu": n_loc, "nnz_per_gpu": nnz, "nnz_total": nnz_glob, "col_idx_bytes": ib, "row_offset_bytes": rb,
                       "x_exchange": lap_mode,
                       "x_exchange_ms_alone": round(ex_ms, 5),
                       "assembly_s": round(asm_s, 4), "assembly_Mtriples_per_s": round(nnz0 / asm_s / 1e6, 1),
                       "settle_steps": SETTLE,
                       "settle_note": "untimed products of the bit-reproducibility gate, run before the warm-up steps"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": round(kern_ms, 5), "algorithmic_bytes": alg_bytes},
        }
        if c5 is not None:
            out["c5"] = c5
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

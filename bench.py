#!/usr/bin/env python3
"""bench.py -- CSR f64 SpMV on the 5-point Laplacian of a 4096^2 grid (BASELINE.json
configs[1]) on N MI355X.

One "step" = one y = A x over the whole matrix (all ranks).  The matrix is built
through the product path itself (shuffled COO -> sr.sparse on the device), x and
the matrix are resident in HBM before the timed region.

N > 1 (one process per GPU, torch.distributed over RCCL): weak scaling.  Rank k
owns the k-th block of 4096*4096 rows of the (N*4096) x 4096 grid's matrix and
the matching slice of x; each step exchanges the x entries other ranks need
(sparse_amd.distributed: halo all-to-all when few columns are remote, all-gather
otherwise), then runs the same SpMV kernel on its rows.

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


def cpu_baseline(seconds=12.0):
    """The oracle's SpMV (a port of compressed.fut:134-146, rows over OpenMP threads)
    timed on this host on a bounded sample: the same stencil on a 2048^2 grid."""
    import numpy as np
    from oracle.oracle import Oracle, lib
    from sparse_amd import synth
    g = 2048
    n = g * g
    r, c, v = synth.laplacian_2d_coo(g)
    order = np.lexsort((c, r))            # the record directly (row-major, columns increasing)
    r, c, v = r[order], c[order], v[order]
    row_off = np.cumsum(np.bincount(r, minlength=n)).astype(np.int64)
    A = {"n": n, "m": n, "nnz": len(r), "row_off": row_off, "col_idx": c, "vals": v}
    x = 1.0 + (np.arange(n) % 7) / 8.0
    threads = min(16, os.cpu_count() or 1) if lib().oracle_has_openmp() else 1
    O = Oracle("f64")
    O.sr_smvm(A, x, threads=threads)      # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        O.sr_smvm(A, x, threads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds or reps >= 400:
            break
    return {"value": round(2.0 * len(r) * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": threads,
            "kind": "port",
            "sample": "5-point Laplacian 2048^2 grid (n=%d, nnz=%d), %d SpMVs, OpenMP row loop of the oracle" % (n, len(r), reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algo", type=int, default=0)
    # rehearsal switches (defaults are the measured configuration)
    ap.add_argument("--grid", type=int, default=GRID, help="grid edge per GPU (default 4096 = BASELINE configs[1])")
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
    t_asm0 = time.perf_counter()
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
            return M.sr.smvm(A, x)
        return D.smvm()

    # ---- parity gate on the exact workload (size-independent property: the stencil itself)
    y = torch.as_tensor(step(), device=dev)
    if world == 1:
        want = synth.laplacian_2d_apply_device(g, x)
    else:
        want = synth.laplacian_2d_apply_rows(g, g * world, rank * n_loc, (rank + 1) * n_loc, dev)
    assert torch.equal(y, want), "SpMV result differs from the stencil applied directly"
    del y, want

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps

    # kernel-only timing (no exchange) for the roofline entry
    if D is not None:
        xf = D.exchange()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.steps):
            D.local_smvm(xf)
        ev1.record()
        torch.cuda.synchronize()
        kern_ms = ev0.elapsed_time(ev1) / args.steps
        ib = ctx.get_option("spmv_last_col_bytes")     # of the dominant kernel (the last product was local_smvm)
        m_loc = D.x_len
        # the x exchange alone (wall clock: the transfers run on the communicator's own stream)
        dist.barrier()
        torch.cuda.synchronize()
        tx = time.perf_counter()
        for _ in range(args.steps):
            D.exchange()
        torch.cuda.synchronize()
        ex_ms = (time.perf_counter() - tx) / args.steps * 1e3
    else:
        kern_ms = dev_ms
        ib = ctx.get_option("spmv_last_col_bytes")
        m_loc = n_loc
        ex_ms = 0.0

    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        tot = torch.tensor([nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        nnz_glob = int(tot.item())
    else:
        nnz_glob = nnz

    if rank == 0:
        ms = el / args.steps * 1e3
        gflops = 2.0 * nnz_glob / (el / args.steps) / 1e9
        # algorithmic bytes of one launch on this rank, as resident on the device and read by the kernel:
        # entries (8 B value + 4 B i32 column index) + row offsets 2 B (tile-relative u16, row_rel) and y 8 B per
        # row + x 8 B per referenced column
        # ib = 2: 16-bit tile-relative columns (+ 4 B per 1024-entry tile)
        ntl = (nnz + 1023) // 1024
        rb = 2 if ntl > 4096 else 8                    # small records (launch-bound path) read row_ptr itself
        alg_bytes = nnz * (8 + ib) + (ntl * 4 if ib == 2 else 0) + n_loc * rb + n_loc * 8 + m_loc * 8
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and g == GRID:   # the PMC pass was taken on the 4096^2 per-GPU workload
            try:
                traffic = json.load(open(tpath)).get("spmv_c2_f64_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CSR f64 SpMV GFLOP/s + achieved HBM GB/s (% peak) at 1/2/4/8 MI355X",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D 5-point Laplacian CSR f64, %d^2 grid per GPU, SpMV (BASELINE configs[1])" % g,
                       "rows_per_gpu": n_loc, "nnz_per_gpu": nnz, "nnz_total": nnz_glob, "col_idx_bytes": ib, "row_offset_bytes": rb,
                       "x_exchange": "none" if D is None else D.mode, "x_exchange_ms_alone": round(ex_ms, 5),
                       "assembly_s": round(asm_s, 4), "assembly_Mtriples_per_s": round(nnz0 / asm_s / 1e6, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms, 5), "algorithmic_bytes": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

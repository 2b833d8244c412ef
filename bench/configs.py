#!/usr/bin/env python3
"""All BASELINE.json configs on ONE MI355X (C5 as its single-GPU shard sizes), with an
independent full-size cross-check against torch's own sparse kernels (rocSPARSE; used
as a comparator only).  Prints one JSON line per config.

    python bench/configs.py [c1 c2 c3 c4 c5] [--steps 50] [--check]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 8000.0


def timed(fn, steps, torch):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c1", "c2", "c3", "c4"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--spmm-algo", type=int, default=0)
    ap.add_argument("--cblk", type=int, default=0, help="spmv_cblk option: -1 never, W > 0 force W columns per block")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--pb", type=int, default=0, help="spmv_pb option: -1 never, 1 also records of flat popularity, 2 always")
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="any context option, key=value (kernel lab)")
    ap.add_argument("--asm-only", action="store_true", help="time the assembly only (kernel labs whose ablations leave a wrong record)")
    ap.add_argument("--gap-ms", type=float, default=0.0, help="lab: synchronise and idle this long between the timed products (spmv_ms then includes it)")
    ap.add_argument("--compare", action="store_true", help="also time torch.sparse (rocSPARSE) CSR SpMV as a comparator")
    args = ap.parse_args()
    import torch
    from sparse_amd import Context, mk_compressed, synth
    dev = torch.device("cuda", 0)
    ctx = Context(0, logging=bool(os.environ.get("SAMD_LOG")))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.algo:
        ctx.set_option("spmv_algo", args.algo)
    if args.cblk:
        ctx.set_option("spmv_cblk", args.cblk)
    if args.pb:
        ctx.set_option("spmv_pb", args.pb)
    if args.spmm_algo:
        ctx.set_option("spmm_algo", args.spmm_algo)
    if args.lanes:
        ctx.set_option("spmv_lanes", args.lanes)
    if args.ablate:
        ctx.set_option("ablate", args.ablate)
    for kv in args.opt:
        key, val = kv.split("=")
        ctx.set_option(key, int(val))

    def run(name, dtype, n, m, gen, spmm_k=0):
        vb = 8 if dtype == "f64" else 4
        tdt = torch.float64 if dtype == "f64" else torch.float32
        M = mk_compressed(dtype, ctx)
        r, c, v = gen()
        nnz0 = r.numel()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A = M.sr.sparse(n, m, (r, c, v))
        ctx.sync()
        asm_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        A2 = M.sr.sparse(n, m, (r, c, v))
        ctx.sync()
        asm = time.perf_counter() - t0
        del A2
        _, _, nnz = A.stored_dims()
        x = torch.rand(m, device=dev, dtype=tdt) + 0.5
        out = {"config": name, "dtype": dtype, "n": n, "m": m, "nnz0": nnz0, "nnz": nnz,
               "assembly_ms": round(asm * 1e3, 3), "assembly_first_ms": round(asm_first * 1e3, 3),
               "assembly_Mtriples_s": round(nnz0 / asm / 1e6, 1),
               "assembly_GBs": round((nnz0 * (16 + vb) + nnz * (4 + vb) + n * 8) / asm / 1e9, 1)}
        if args.asm_only:
            print(json.dumps(out), flush=True)
            del A, r, c, v
            ctx.clear_caches()
            torch.cuda.empty_cache()
            return
        if args.check:
            ref = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, m)).coalesce()
            ro, ci, va = A.fields()
            ci_t = torch.as_tensor(ci, device=dev)
            va_t = torch.as_tensor(va, device=dev)
            ro_t = torch.as_tensor(ro, device=dev)
            idx = ref.indices()
            out["check_cols_equal"] = bool(torch.equal(idx[1], ci_t))
            out["check_rowoff_equal"] = bool(torch.equal(torch.cumsum(torch.bincount(idx[0], minlength=n), 0), ro_t))
            out["check_vals_maxrel"] = float(((ref.values() - va_t).abs() / ref.values().abs()).max())
            yref = torch.sparse.mm(ref.to_sparse_csr(), x[:, None])[:, 0]
            y = torch.as_tensor(M.sr.smvm(A, x), device=dev)
            out["check_smvm_maxrel"] = float(((y - yref).abs() / yref.abs().clamp_min(1e-30)).max())
            del ref, yref, y, idx, ci_t, va_t, ro_t, ro, ci, va
        del r, c, v
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.sr.smvm(A, x)        # a record's first product runs on what the assembly left (nothing built)
        ctx.sync()
        out["spmv_first_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        t0 = time.perf_counter()
        M.sr.prepare(A, 1 << 20)   # "many products follow": whatever the record's path needs is built here, once
        ctx.sync()
        out["spmv_prepare_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        def product():
            M.sr.smvm(A, x)
            if args.gap_ms > 0:
                torch.cuda.synchronize()
                time.sleep(args.gap_ms * 1e-3)
        ms = timed(product, args.steps, torch)
        # row offsets as the kernel reads them: 2 B (tile-relative u16) on the one-pass stream path of large records
        rb = 2 if (ctx.get_option("spmv_last_path") == 0 and nnz > 4096 * 1024 and not args.algo) else 8
        ib = ctx.get_option("spmv_last_col_bytes") if ctx.get_option("spmv_last_path") == 0 else 4
        byt = nnz * (vb + ib) + n * rb + n * vb + m * vb
        out.update({"spmv_ms": round(ms, 4), "spmv_GFLOPs": round(2 * nnz / ms / 1e6, 1),
                    "spmv_GBs": round(byt / ms / 1e6, 1), "spmv_frac_peak": round(byt / ms / 1e6 / PEAK, 4)})
        out["spmv_path_id"] = ctx.get_option("spmv_last_path")
        if ctx.get_option("spmv_last_path") == 3:
            out["spmv_path"] = "two passes"
            out["pb_fold"] = ctx.get_option("spmv_last_pb_fold")
            out["pb_products"] = ctx.get_option("spmv_last_pb_outputs")
        if args.compare:
            r2, c2, v2 = M.sr.coo(A)
            ref = torch.sparse_coo_tensor(torch.stack([torch.as_tensor(r2, device=dev), torch.as_tensor(c2, device=dev)]),
                                          torch.as_tensor(v2, device=dev), (n, m)).coalesce().to_sparse_csr()
            del r2, c2, v2
            xm = x[:, None].contiguous()
            ms_ref = timed(lambda: torch.sparse.mm(ref, xm), max(5, args.steps // 5), torch)
            out["rocsparse_via_torch_spmv_ms"] = round(ms_ref, 4)
            del ref
            torch.cuda.empty_cache()
        if spmm_k:
            B = torch.rand(m, spmm_k, device=dev, dtype=tdt) + 0.5
            ms = timed(lambda: M.sr.smm(A, B), max(5, args.steps // 5), torch)
            byt = nnz * (vb + 4) + n * 8 + (m + n) * spmm_k * vb
            out.update({"spmm_k": spmm_k, "spmm_ms": round(ms, 4), "spmm_GFLOPs": round(2 * nnz * spmm_k / ms / 1e6, 1),
                        "spmm_GBs": round(byt / ms / 1e6, 1), "spmm_frac_peak": round(byt / ms / 1e6 / PEAK, 4)})
            if args.check:
                C = torch.as_tensor(M.sr.smm(A, B), device=dev)
                r2, c2, v2 = M.sr.coo(A)
                ref = torch.sparse_coo_tensor(torch.stack([torch.as_tensor(r2, device=dev), torch.as_tensor(c2, device=dev)]),
                                              torch.as_tensor(v2, device=dev), (n, m)).to_sparse_csr()
                Cref = torch.sparse.mm(ref, B)
                out["check_spmm_maxrel"] = float(((C - Cref).abs() / Cref.abs().clamp_min(1e-30)).max())
            t0 = None
        # physical conversion
        ms = timed(lambda: M.sr.to_sc(A), 3, torch)
        out["to_csc_ms"] = round(ms, 3)
        out["to_csc_GBs"] = round((2 * nnz * (4 + vb) + (n + m) * 8) / ms / 1e6, 1)
        print(json.dumps(out), flush=True)
        del A
        ctx.clear_caches()
        torch.cuda.empty_cache()

    for cfg in args.configs:
        if cfg == "c1":
            run("C1 random 1e5x1e5 nnz0=1e6", "f64", 100000, 100000,
                lambda: synth.random_uniform_coo_device(100000, 100000, 1000000, dev))
        elif cfg == "c2":
            g = 4096
            run("C2 laplacian 4096^2", "f64", g * g, g * g, lambda: synth.laplacian_2d_coo_device(g, dev))
        elif cfg == "c3":
            run("C3 RMAT scale 24 ef 18", "f32", 1 << 24, 1 << 24, lambda: synth.rmat_coo_device(24, 18, dev, "f32"))
        elif cfg == "c3d":   # the same graph with 8-byte values
            run("lab: RMAT scale 24 ef 18, f64", "f64", 1 << 24, 1 << 24, lambda: synth.rmat_coo_device(24, 18, dev, "f64"))
        elif cfg == "c3s":
            run("C3-small RMAT scale 20 ef 18", "f32", 1 << 20, 1 << 20, lambda: synth.rmat_coo_device(20, 18, dev, "f32"))
        elif cfg == "c4":
            n = 1 << 20
            run("C4 banded n=2^20 hb=13, k=64", "f64", n, n, lambda: synth.banded_coo_device(n, 13, dev), spmm_k=64)
        elif cfg == "c5":
            run("C5 random n=1e7 nnz0=1e9 (one GPU)", "f64", 10**7, 10**7,
                lambda: synth.random_uniform_coo_device(10**7, 10**7, 10**9, dev))
        elif cfg == "c5r":   # what one of 8 ranks holds after row sharding: an eighth of the rows, every column
            run("C5 row shard 1/8: random 1.25e6 x 1e7, nnz0=1.25e8", "f64", 1250000, 10**7,
                lambda: synth.random_uniform_coo_device(1250000, 10**7, 125 * 10**6, dev))
        elif cfg == "c5x":   # SURVEY 8(d)'s stress configuration (n = m = 1e8, ~10 per row) as one of 8 row shards
            run("C5 stress shard 1/8: random 1.25e7 x 1e8, nnz0=1.25e8", "f64", 12500000, 10**8,
                lambda: synth.random_uniform_coo_device(12500000, 10**8, 125 * 10**6, dev))
        elif cfg == "c5q":   # kernel lab: few enough panels for one round of resident workgroups
            run("lab: random 5e5 x 1e7, nnz0=5e7", "f64", 500000, 10**7,
                lambda: synth.random_uniform_coo_device(500000, 10**7, 50 * 10**6, dev))
        elif cfg == "c5s":
            run("C5-shard random n=1e7 nnz0=1.25e8", "f64", 10**7, 10**7,
                lambda: synth.random_uniform_coo_device(10**7, 10**7, 125 * 10**6, dev))


if __name__ == "__main__":
    main()

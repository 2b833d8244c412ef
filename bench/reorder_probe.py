#!/usr/bin/env python3
"""Probe: does renumbering the columns of a power-law matrix by popularity (hot columns packed into
the same cache lines) speed up the x gather of SpMV?  Experiment only; prints timings."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparse_amd import Context, mk_compressed, synth

def timed(fn, steps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
M = mk_compressed("f32", ctx)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << scale
r, c, v = synth.rmat_coo_device(scale, 18, dev, "f32")
x = torch.rand(n, device=dev, dtype=torch.float32) + 0.5
A = M.sr.sparse(n, n, (r, c, v))
t0 = timed(lambda: M.sr.smvm(A, x))
y0 = torch.as_tensor(M.sr.smvm(A, x), device=dev).clone()
del A
cnt = torch.bincount(c, minlength=n)
order = torch.argsort(cnt, descending=True, stable=True)      # order[rank] = column
rank = torch.empty_like(order); rank[order] = torch.arange(n, device=dev)
c2 = rank[c]
A2 = M.sr.sparse(n, n, (r, c2, v))
xp = x[order].contiguous()
t1 = timed(lambda: M.sr.smvm(A2, xp))
y1 = torch.as_tensor(M.sr.smvm(A2, xp), device=dev)
err = float(((y1 - y0).abs() / y0.abs().clamp_min(1e-30)).max())
del A2
# rows too (symmetric renumbering by row degree)
rc = torch.bincount(r, minlength=n)
rorder = torch.argsort(rc, descending=True, stable=True); rrank = torch.empty_like(rorder); rrank[rorder] = torch.arange(n, device=dev)
A3 = M.sr.sparse(n, n, (rrank[r], c2, v))
t2 = timed(lambda: M.sr.smvm(A3, xp))
top = int(cnt[order[: (1 << 20)]].sum()) / int(cnt.sum())
print(json.dumps({"scale": scale, "spmv_ms": round(t0, 4), "cols_by_popularity_ms": round(t1, 4), "rows_and_cols_ms": round(t2, 4),
                  "maxrel_vs_plain": err, "mass_top_1M_cols": round(top, 3)}))

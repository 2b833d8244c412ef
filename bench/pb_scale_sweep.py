#!/usr/bin/env python3
"""Kernel lab: two-pass SpMV against the one-pass paths on RMAT matrices of several scales (f32, i32)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_amd import Context, mk_compressed, synth


def timed(fn, steps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for scale in [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23]:
    n = 1 << scale
    r, c, v = synth.rmat_coo_device(scale, 18, "cuda:0", "f32")
    x = torch.rand(n, device="cuda:0", dtype=torch.float32) + 0.5
    M = mk_compressed("f32", ctx)
    row = {"scale": scale}
    for name, pb in (("auto", 0), ("one_pass", -1), ("two_pass_forced", 2)):
        ctx.set_option("spmv_pb", pb)
        A = M.sr.sparse(n, n, (r, c, v))
        row[name + "_ms"] = round(timed(lambda: M.sr.smvm(A, x)), 4)
        row[name + "_path"] = ctx.get_option("spmv_last_path")
        A.free()
    ctx.set_option("spmv_pb", 0)
    print(json.dumps(row), flush=True)

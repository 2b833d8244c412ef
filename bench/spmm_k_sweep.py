"""SpMM on the C4 matrix for several panel widths, matrix-core path (default) against spmm_algo 3 (entry path)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_amd import Context, mk_compressed, synth
from configs import timed

dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for dtype, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    M = mk_compressed(dtype, ctx)
    n = 1 << 20
    r, c, v = synth.banded_coo_device(n, 13, dev, dtype)
    A = M.sr.sparse(n, n, (r, c, v))
    for k in (8, 16, 32, 64, 128, 256):
        B = torch.rand(n, k, device=dev, dtype=tdt) + 0.5
        out = {"dtype": dtype, "k": k}
        for algo in (0, 3):
            ctx.set_option("spmm_algo", algo)
            out["ms_algo%d" % algo] = round(timed(lambda: M.sr.smm(A, B), 20, torch), 4)
        ctx.set_option("spmm_algo", 0)
        print(json.dumps(out), flush=True)

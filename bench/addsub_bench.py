"""sr.+ on two records of C2's size (5-point Laplacian 4096^2, f64) and C4's: the merge of the two key-ordered halves
against the sort of their concatenation (option add_sort).  One JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from sparse_amd import Context, mk_compressed, synth


def timed(f, reps):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    M = mk_compressed("f64", ctx)
    for name, n, gen in (("C2 laplacian 4096^2", 4096 * 4096, lambda: synth.laplacian_2d_coo_device(4096, dev)),
                         ("C4 banded 2^20 hb=13", 1 << 20, lambda: synth.banded_coo_device(1 << 20, 13, dev))):
        r, c, v = gen()
        A = M.sr.sparse(n, n, (r, c, v))
        B = M.sr.sparse(n, n, (c, r, v))          # the transposed pattern: same size, partly different coordinates
        del r, c, v
        out = {"case": name, "nnz_a": A.stored_dims()[2], "nnz_b": B.stored_dims()[2]}
        for key, opt in (("merge_ms", 0), ("sort_ms", 1)):
            ctx.set_option("add_sort", opt)
            out[key] = round(timed(lambda: M.sr.add(A, B), 5), 3)
        ctx.set_option("add_sort", 0)
        out["nnz_sum"] = M.sr.add(A, B).stored_dims()[2]
        print(json.dumps(out), flush=True)
        del A, B
        ctx.clear_caches()


if __name__ == "__main__":
    main()

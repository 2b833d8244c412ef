"""One rank of a 2-rank gloo world whose ranks share cuda:0 (launched by
tests/test_zz_gpu_dist_rehearsal.py).  Distributed assembly, SpMV and SpMM through DistCsr
against the single-GPU product path on the concatenated input."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sparse_amd import Context, mk_compressed, synth  # noqa: E402
from sparse_amd.distributed import DistCsr, HostStagedDist  # noqa: E402


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hd = HostStagedDist(dist)      # ranks share one GPU: collectives through CPU copies (gloo + CUDA tensors hung)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    M = mk_compressed("f64", ctx)

    def chunk(a):
        cuts = [0] + [(len(a) * (k + 1) * (k + 2)) // (world * (world + 1)) for k in range(world)]
        return a[cuts[rank]:cuts[rank + 1]]

    # ---- random matrix with duplicates; integer-valued f64 so that every sum is exact
    n, k = 20000, 400000
    idx = np.arange(k, dtype=np.uint64)
    r = (synth.splitmix64(21, idx) % np.uint64(n)).astype(np.int64)
    c = (synth.splitmix64(22, idx) % np.uint64(n // 8)).astype(np.int64) * 8
    v = (synth.splitmix64(23, idx) % np.uint64(9)).astype(np.float64) - 4.0
    G = M.sr.sparse(n, n, (r, c, v))
    g_ro, g_ci, g_va = (np.asarray(a) for a in G.fields())
    x = 1.0 + (np.arange(n) % 7).astype(np.float64)
    B = (np.arange(n * 8) % 5).astype(np.float64).reshape(n, 8) - 2.0
    y_ref = np.asarray(M.sr.smvm(G, x))
    C_ref = np.asarray(M.sr.smm(G, B))

    # distributed assembly, global column numbering, one record per rank: bit-equal row blocks
    D = DistCsr.from_coo_chunks(M, hd, n, n, t(chunk(r), dev), t(chunk(c), dev), t(chunk(v), dev),
                                force_mode="allgather", split=False)
    lo, hi = D.bounds[rank], D.bounds[rank + 1]
    off = D.nnz_offset()
    l_ro, l_ci, l_va = (np.asarray(a) for a in D.A.fields())
    assert np.array_equal(l_ro + off, g_ro[lo:hi]), "row_off"
    assert np.array_equal(l_ci, g_ci[off:off + len(l_ci)]), "col_idx"
    assert l_va.tobytes() == g_va[off:off + len(l_va)].tobytes(), "vals"
    for mode in ("allgather", "halo", None):
        D = DistCsr.from_coo_chunks(M, hd, n, n, t(chunk(r), dev), t(chunk(c), dev), t(chunk(v), dev),
                                    force_mode=mode)
        lo, hi = D.bounds[rank], D.bounds[rank + 1]
        y = D.smvm(t(x[lo:hi], dev)).cpu().numpy()
        assert np.array_equal(y, y_ref[lo:hi]), ("smvm", mode)
        C = np.asarray(D.smm(t(B[lo:hi], dev)))
        assert np.array_equal(C, C_ref[lo:hi]), ("smm", mode)

    # ---- banded matrix: halo mode by itself, split record, overlap path
    rb, cb, vb = synth.banded_coo(4096, 5, "f64", seed=5)
    vb = np.round(vb * 8.0)
    n2 = 4096
    G2 = M.sr.sparse(n2, n2, (rb, cb, vb))
    x2 = 1.0 + (np.arange(n2) % 5).astype(np.float64)
    y2 = np.asarray(M.sr.smvm(G2, x2))
    perm = synth.splitmix64(31, np.arange(len(rb), dtype=np.uint64)).argsort()
    D2 = DistCsr.from_coo_chunks(M, hd, n2, n2, t(chunk(rb[perm]), dev), t(chunk(cb[perm]), dev),
                                 t(chunk(vb[perm]), dev))
    assert D2.mode == "halo" and D2.A_gh is not None
    lo, hi = D2.bounds[rank], D2.bounds[rank + 1]
    for rep in range(3):
        y = D2.smvm(t((rep + 1) * x2[lo:hi], dev)).cpu().numpy()
        assert np.array_equal(y, (rep + 1) * y2[lo:hi]), ("banded smvm", rep)
    # ---- rectangular matrix (x cut into equal slices, rows by the triple histogram): both exchange modes
    n3, m3, k3 = 9000, 5000, 120000
    idx = np.arange(k3, dtype=np.uint64)
    r3 = (synth.splitmix64(41, idx) % np.uint64(n3)).astype(np.int64)
    c3 = (synth.splitmix64(42, idx) % np.uint64(m3)).astype(np.int64)
    v3 = (synth.splitmix64(43, idx) % np.uint64(9)).astype(np.float64) - 4.0
    G3 = M.sr.sparse(n3, m3, (r3, c3, v3))
    x3 = 1.0 + (np.arange(m3) % 7).astype(np.float64)
    y3 = np.asarray(M.sr.smvm(G3, x3))
    for mode in ("allgather", "halo"):
        D3 = DistCsr.from_coo_chunks(M, hd, n3, m3, t(chunk(r3), dev), t(chunk(c3), dev), t(chunk(v3), dev),
                                     force_mode=mode)
        lo, hi = D3.bounds[rank], D3.bounds[rank + 1]
        xlo, xhi = D3.plan.lo, D3.plan.hi
        y = D3.smvm(t(x3[xlo:xhi], dev)).cpu().numpy()
        assert np.array_equal(y, y3[lo:hi]), ("rectangular smvm", mode)
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

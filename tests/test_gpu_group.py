"""The native multi-GPU API (include/sparse_amd.h section 6) through the C ABI.  One GPU is all a test box has,
so several members share device 0 over the peer transport (direct copies ordered by events -- the same code
path distinct devices take when RCCL is not asked for); the RCCL transport itself needs distinct devices, so what
a one-GPU box can run of it is a one-member group named "rccl": librccl opened, ncclCommInitAll, and the member's
slice of x moved by a grouped ncclSend / ncclRecv to itself (test_rccl_transport_runs_on_one_member).  Checked: the members' records are, bit for bit, the row
blocks of the oracle's single record (duplicates summed in input order); y = A x against the oracle; a chain of
products whose result is the next operand (square matrices: x is cut like the rows)."""
import numpy as np
import pytest

from helpers import random_coo
from oracle.oracle import Oracle
from sparse_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_assembly_and_smvm_match_the_oracle(members):
    from sparse_amd.group import Group
    g = Group([0] * members, "peer" if members > 1 else None)
    assert g.transport == ("peer" if members > 1 else "none")
    O = Oracle("f64")
    for n, m, k, seed in ((5000, 5000, 120000, 3), (9000, 4000, 60000, 4), (64, 70000, 30000, 5)):
        r, c, v = random_coo(n, m, k, "f64", seed=seed, dup_frac=0.1)
        Ao = O.sr_sparse(n, m, r, c, v)
        A = g.sparse(n, m, (r, c, v))
        rb, cb = A.bounds()
        assert rb[0] == 0 and rb[-1] == n and cb[0] == 0 and cb[-1] == m and all(a <= b for a, b in zip(rb, rb[1:]))
        off = 0
        for q in range(members):
            ro, ci, va = A.member_fields(q)
            lo, hi = rb[q], rb[q + 1]
            assert np.array_equal(ro + off, Ao["row_off"][lo:hi])
            assert np.array_equal(ci, Ao["col_idx"][off:off + len(ci)])
            assert va.tobytes() == Ao["vals"][off:off + len(va)].tobytes()
            off += len(ci)
        assert off == Ao["nnz"]
        x = synth.uniform_values(seed + 10, m, "f64")
        xv, yv = g.vector(A, x), g.vector(A, None, like_rows=True)
        for rep in range(3):                              # the x buffers are reused: later products wait for earlier ones
            g.smvm(yv, A, xv)
        np.testing.assert_allclose(yv.numpy(), O.sr_smvm(Ao, x), rtol=1e-12, atol=0)
        if n == m:                                        # y feeds the next product
            xi = np.round(x * 4)                          # small integers: every sum exact
            ri, ci_, vi = r, c, np.round(v * 4)
            Ai, Aio = g.sparse(n, m, (ri, ci_, vi)), O.sr_sparse(n, m, ri, ci_, vi)
            a, b = g.vector(Ai, xi), g.vector(Ai, None, like_rows=True)
            g.smvm(b, Ai, a)
            g.smvm(a, Ai, b)
            want = O.sr_smvm(Aio, O.sr_smvm(Aio, xi))
            assert np.array_equal(a.numpy(), want)
            for o in (a, b, Ai):
                o.free()
        for o in (xv, yv, A):
            o.free()
    g.close()


def test_group_errors_and_transport_choice():
    from sparse_amd.compressed import SparseError
    from sparse_amd.group import Group
    with pytest.raises(SparseError, match="distinct"):
        Group([0, 0], "rccl")
    with pytest.raises(SparseError, match="transport"):
        Group([0], "carrier-pigeon")
    with pytest.raises(SparseError, match="members"):     # the group object still comes back, with the text in it
        Group([], None)
    g = Group([0], None)                                  # one member, nothing asked for: librccl is not even opened
    assert g.transport == "none"
    with pytest.raises(SparseError, match="out of bounds"):
        g.sparse(3, 3, (np.array([3]), np.array([0]), np.array([1.0])))
    A = g.sparse(4, 3, (np.array([0, 3]), np.array([1, 2]), np.array([2.0, 5.0])))
    x, y = g.vector(A, np.array([1.0, 10.0, 100.0])), g.vector(A, None, like_rows=True)
    bad = g.vector(A, None, like_rows=True)
    with pytest.raises(SparseError, match="partitioned"):
        g.smvm(y, A, bad)
    g.smvm(y, A, x)
    assert np.array_equal(y.numpy(), [20.0, 0.0, 0.0, 500.0])
    g.close()


def test_c_caller_shards_without_python():
    """integration/group_example.c: a plain C program over the group API (two members rehearsed on device 0)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "integration", "group_example")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    for args in (["1"], ["2", "peer"]):
        out = subprocess.run([exe] + args, capture_output=True, timeout=100)
        assert out.returncode == 0, out.stderr.decode()
        assert b"group ok" in out.stdout


def test_rccl_transport_runs_on_one_member():
    """group.hpp's RCCL path end to end on the one GPU there is: dlopen(librccl), ncclCommInitAll over one device,
    and every product's x slice travelling through ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd to itself
    on the member's stream, followed by the ordinary product.  y against the oracle; x_full is overwritten between
    products so a transfer that silently did nothing would show."""
    from sparse_amd.group import Group
    g = Group([0], "rccl")
    assert g.transport == "rccl" and g.size == 1
    O = Oracle("f64")
    n, m, k = 6000, 7000, 150000
    r, c, v = random_coo(n, m, k, "f64", seed=21, dup_frac=0.05)
    Ao = O.sr_sparse(n, m, r, c, v)
    A = g.sparse(n, m, (r, c, v))
    yv = g.vector(A, None, like_rows=True)
    for seed in (31, 32, 33):
        x = synth.uniform_values(seed, m, "f64")
        xv = g.vector(A, x)
        g.smvm(yv, A, xv)
        np.testing.assert_allclose(yv.numpy(), O.sr_smvm(Ao, x), rtol=1e-12, atol=0)
        xv.free()
    for o in (yv, A):
        o.free()
    g.close()

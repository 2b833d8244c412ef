"""World-size-2 (and 3) gloo runs of the row-sharding logic on the CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_plan_gloo(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
    assert "DIST_OK world=%d" % world in outs[0]


def test_partition_rows_equal_nnz():
    from sparse_amd.distributed import partition_rows_equal_nnz
    lens = np.array([5, 0, 0, 7, 1, 1, 1, 30, 2, 2])
    ro = np.cumsum(lens)
    b = partition_rows_equal_nnz(ro, 4)
    assert b[0] == 0 and b[-1] == len(lens) and all(x <= y for x, y in zip(b, b[1:]))
    # single part, empty matrix, more parts than rows
    assert partition_rows_equal_nnz(ro, 1) == [0, len(lens)]
    assert partition_rows_equal_nnz(np.zeros(0, np.int64), 3) == [0, 0, 0, 0]
    assert partition_rows_equal_nnz(np.array([4]), 3)[-1] == 1
    # uniform rows split evenly
    assert partition_rows_equal_nnz(np.cumsum(np.full(100, 5)), 4) == [0, 25, 50, 75, 100]

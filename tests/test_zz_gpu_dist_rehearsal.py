"""Two ranks on ONE GPU over gloo: the whole N>1 bench path (row shards, ExchangePlan in
halo and all-gather mode, DistCsr split product, parity gate against the stencil) run end to
end.  Throughput is meaningless here; bench.py asserts y == stencil bit for bit on every rank
before it times anything, so exit status 0 + a JSON line is the check."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, ranks=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--grid", "512", "--backend", "gloo",
           "--same-device", "--watchdog", "100", "--c5-n", "20000", "--c5-triples", "400000"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_halo():
    j = _run([], 29541)
    assert j["n_gpus"] == 2 and j["config"]["x_exchange"] == "halo"
    # the strong-scaling leg (BASELINE configs[4] at rehearsal size): distributed assembly from per-rank slices
    # of the counter-PRNG triple list, all-gather of x, parity gate inside bench.py
    assert j["c5"]["n_gpus"] == 2 and j["c5"]["x_exchange"] == "allgather" and 390000 < j["c5"]["nnz_total"] <= 400000
    assert j["config"]["nnz_total"] == 5 * 512 * 1024 - 2 * 512 - 2 * 1024      # one 512 x 1024 grid


@pytest.mark.gpu
def test_eight_ranks_halo_and_the_strong_scaling_leg():
    """The driver's largest run in miniature: 8 ranks (all on one GPU here), interior ranks with two halo neighbours,
    the 1e9-triple leg at rehearsal size cut into 8 row shards."""
    j = _run([], 29544, ranks=8)
    assert j["n_gpus"] == 8 and j["config"]["x_exchange"] == "halo"
    assert j["c5"]["n_gpus"] == 8 and j["c5"]["x_exchange"] == "allgather" and 390000 < j["c5"]["nnz_total"] <= 400000
    assert j["config"]["nnz_total"] == 5 * 512 * 4096 - 2 * 512 - 2 * 4096      # one 512 x 4096 grid


@pytest.mark.gpu
def test_two_ranks_allgather():
    j = _run(["--exchange", "allgather"], 29542)
    assert j["n_gpus"] == 2 and j["config"]["x_exchange"] == "allgather"


@pytest.mark.gpu
def test_two_ranks_assembly_smvm_smm():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and "DIST_GPU_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

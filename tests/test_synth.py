"""The device-side counter PRNG (torch, int64 bit patterns) agrees bit for bit with the host one (numpy
uint64): any rank can generate any slice of the same global triple list (bench.py's strong-scaling leg)."""
import numpy as np
import torch

from sparse_amd import synth


def test_splitmix64_device_matches_host():
    for seed in (0, 42, 2**63 + 5):
        want = synth.splitmix64(seed, np.arange(1000, 1400, dtype=np.uint64))
        got = synth.splitmix64_device(seed, 1000, 400, "cpu").numpy().view(np.uint64)
        assert np.array_equal(got, want)


def test_counter_coo_slices_match_the_host_generator():
    n, m = 10**7, 10**7 - 3
    r, c, v = synth.random_uniform_coo(n, m, 6000, "f64", seed=42)
    for start, count in ((0, 6000), (1234, 777), (5999, 1)):
        r2, c2, v2 = synth.random_uniform_coo_counter_device(n, m, start, count, "cpu")
        assert np.array_equal(r2.numpy(), r[start:start + count])
        assert np.array_equal(c2.numpy(), c[start:start + count])
        assert np.array_equal(v2.numpy(), v[start:start + count])
    _, _, vi = synth.random_uniform_coo_counter_device(n, m, 0, 500, "cpu", integer_values=True)
    want = (synth.splitmix64(44, np.arange(500, dtype=np.uint64)) % np.uint64(19)).astype(np.int64) - 9
    assert np.array_equal(vi.numpy(), want.astype(np.float64))
    # the slices of any partition concatenate to the whole list
    parts = [synth.random_uniform_coo_counter_device(n, m, 6000 * k // 3, 6000 * (k + 1) // 3 - 6000 * k // 3, "cpu")
             for k in range(3)]
    assert torch.equal(torch.cat([p[0] for p in parts]), torch.from_numpy(r))

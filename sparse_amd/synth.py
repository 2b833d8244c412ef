"""Seeded synthetic inputs (host side, numpy).

Everything derives from splitmix64 on (seed, index), a counter-based generator,
so any slice of any input can be regenerated independently (per-rank shards in
the multi-GPU bench) and the C twin in oracle/compressed_oracle.c agrees bit for
bit.  Shapes follow BASELINE.json's configs.
"""
import numpy as np

_U64 = np.uint64


def splitmix64(seed, index):
    """Vectorised splitmix64 finaliser of seed + (index+1)*golden."""
    with np.errstate(over="ignore"):
        idx = np.asarray(index, dtype=np.uint64)
        z = _U64(seed & 0xFFFFFFFFFFFFFFFF) + (idx + _U64(1)) * _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def uniform01(seed, count, start=0):
    """f64 uniform in [0,1) from the top 53 bits."""
    z = splitmix64(seed, np.arange(start, start + count, dtype=np.uint64))
    return (z >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform_values(seed, count, dtype, start=0):
    """Matrix/vector values: floats uniform in [0.5,1.5) (positive, so relative
    tolerances are meaningful); i32 uniform in [-9,9] (exact arithmetic)."""
    if dtype == "i32":
        z = splitmix64(seed, np.arange(start, start + count, dtype=np.uint64))
        return ((z % _U64(19)).astype(np.int64) - 9).astype(np.int32)
    u = uniform01(seed, count, start) + 0.5
    return u.astype(np.float64 if dtype == "f64" else np.float32)


def random_uniform_coo(n, m, nnz0, dtype, seed=42):
    """C1/C5 shape: uniform random (r,c) triples; duplicates occur naturally."""
    i = np.arange(nnz0, dtype=np.uint64)
    r = (splitmix64(seed, i) % _U64(n)).astype(np.int64)
    c = (splitmix64(seed + 1, i) % _U64(m)).astype(np.int64)
    return r, c, uniform_values(seed + 2, nnz0, dtype)


def laplacian_2d_coo(g, dtype="f64", row_lo=0, row_hi=None):
    """C2 shape: 5-point stencil on a g x g grid, rows [row_lo,row_hi) of the
    n=g*g matrix, emitted row-major: (i,i)=4, (i,i+-1)=-1 in-row, (i,i+-g)=-1."""
    n = g * g
    row_hi = n if row_hi is None else row_hi
    i = np.arange(row_lo, row_hi, dtype=np.int64)
    gx = i % g
    gy = i // g
    rows = [i, i[gx > 0], i[gx < g - 1], i[gy > 0], i[gy < g - 1]]
    cols = [i, i[gx > 0] - 1, i[gx < g - 1] + 1, i[gy > 0] - g, i[gy < g - 1] + g]
    vals = [np.full(len(rows[0]), 4)] + [np.full(len(r), -1) for r in rows[1:]]
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    v = np.concatenate(vals).astype({"f64": np.float64, "f32": np.float32, "i32": np.int32}[dtype])
    return r, c, v


def affine_shuffle(count, seed=42):
    """A cheap deterministic permutation i -> (a*i + b) mod count, a coprime to count."""
    if count <= 1:
        return np.arange(count, dtype=np.int64)
    a = int(splitmix64(seed, 0) % _U64(count)) | 1
    while np.gcd(a, count) != 1:
        a += 2
    b = int(splitmix64(seed, 1) % _U64(count))
    i = np.arange(count, dtype=np.uint64)
    # a*i can exceed 2^64 for count > 2^32; counts used here are < 2^32
    return ((i * _U64(a % count) + _U64(b)) % _U64(count)).astype(np.int64)


def banded_coo(n, half_bw, dtype, seed=42):
    """C4 shape: banded n x n, 2*half_bw+1 entries per row clipped at the edges."""
    i = np.arange(n, dtype=np.int64)
    rs, cs = [], []
    for d in range(-half_bw, half_bw + 1):
        ok = (i + d >= 0) & (i + d < n)
        rs.append(i[ok])
        cs.append(i[ok] + d)
    r = np.concatenate(rs)
    c = np.concatenate(cs)
    return r, c, uniform_values(seed, len(r), dtype)


def rmat_coo(scale, edge_factor, dtype, seed=42, probs=(0.57, 0.19, 0.19, 0.05), start=0, count=None):
    """C3 shape: RMAT edges; edge e draws one quadrant per bit level from
    splitmix64(seed+level, e).  Duplicates are left for assembly to merge."""
    n = 1 << scale
    total = n * edge_factor
    count = total - start if count is None else count
    e = np.arange(start, start + count, dtype=np.uint64)
    r = np.zeros(count, np.int64)
    c = np.zeros(count, np.int64)
    a, b, cc, _ = probs
    for lvl in range(scale):
        u = (splitmix64(seed + 101 * (lvl + 1), e) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))
        rbit = (u >= a + b).astype(np.int64)
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + cc)).astype(np.int64)
        r = (r << 1) | rbit
        c = (c << 1) | cbit
    return r, c, uniform_values(seed + 7, count, dtype, start)


# ---------------------------------------------------------------- device side
# Same shapes generated directly in HBM with torch (plumbing only: device
# memory and index arithmetic), for inputs too large to ship over PCIe quickly.
def _affine_perm_device(cnt, device):
    import torch
    a = int(splitmix64(42, 0) % _U64(cnt)) | 1
    while np.gcd(a, cnt) != 1:
        a += 2
    a %= cnt
    b = int(splitmix64(42, 1) % _U64(cnt))
    assert cnt < (1 << 31)  # a*k stays inside int64
    return (torch.arange(cnt, dtype=torch.int64, device=device) * a + b) % cnt


def laplacian_2d_coo_device(g, device, dtype="f64", shuffle=True, row_lo=0, row_hi=None, gy=None):
    """5-point stencil on a gy x g grid (gy = g by default: the C2 shape), rows
    [row_lo,row_hi) of the n = gy*g matrix, generated on the device; entries
    optionally permuted by an affine map so assembly sees unsorted input.
    Returns (rs, cs, vs) torch tensors with GLOBAL indices."""
    import torch
    gy = g if gy is None else gy
    n = gy * g
    row_hi = n if row_hi is None else row_hi
    i = torch.arange(row_lo, row_hi, dtype=torch.int64, device=device)
    x = i % g
    y = i // g
    tdt = {"f64": torch.float64, "f32": torch.float32, "i32": torch.int32}[dtype]
    masks = [None, x > 0, x < g - 1, y > 0, y < gy - 1]
    offs = [0, -1, 1, -g, g]
    rs, cs, vs = [], [], []
    for msk, off in zip(masks, offs):
        r = i if msk is None else i[msk]
        rs.append(r)
        cs.append(r + off)
        vs.append(torch.full((r.numel(),), 4 if off == 0 else -1, dtype=tdt, device=device))
    r = torch.cat(rs)
    c = torch.cat(cs)
    v = torch.cat(vs)
    del rs, cs, vs
    if shuffle:
        p = _affine_perm_device(r.numel(), device)
        r, c, v = r[p], c[p], v[p]
    return r.contiguous(), c.contiguous(), v.contiguous()


def laplacian_x_device(lo, hi, device):
    """x_i = 1 + (i mod 7)/8 for global indices [lo,hi): exactly representable, so
    every stencil sum is exact in f64 whatever the summation order."""
    import torch
    return 1.0 + (torch.arange(lo, hi, device=device) % 7).double() / 8.0


def laplacian_2d_apply_rows(g, gy, row_lo, row_hi, device):
    """(A x)[row_lo:row_hi] for the stencil on a gy x g grid and x = laplacian_x_device,
    computed from the formula alone (no matrix, no communication)."""
    import torch
    i = torch.arange(row_lo, row_hi, dtype=torch.int64, device=device)
    xv = lambda j: 1.0 + (j % 7).double() / 8.0
    x = i % g
    y = i // g
    out = 4.0 * xv(i)
    out -= torch.where(x > 0, xv(i - 1), torch.zeros((), dtype=torch.float64, device=device))
    out -= torch.where(x < g - 1, xv(i + 1), torch.zeros((), dtype=torch.float64, device=device))
    out -= torch.where(y > 0, xv(i - g), torch.zeros((), dtype=torch.float64, device=device))
    out -= torch.where(y < gy - 1, xv(i + g), torch.zeros((), dtype=torch.float64, device=device))
    return out


def laplacian_2d_apply_device(g, x):
    """y = A x for the square 5-point stencil, computed without the matrix (property check)."""
    X = x.reshape(g, g)
    Y = 4 * X
    Y[:, 1:] -= X[:, :-1]
    Y[:, :-1] -= X[:, 1:]
    Y[1:, :] -= X[:-1, :]
    Y[:-1, :] -= X[1:, :]
    return Y.reshape(-1)


def random_uniform_coo_device(n, m, nnz0, device, dtype="f64", seed=42):
    """C1/C5 shape on the device (torch's Philox generator; not the splitmix stream)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    r = torch.randint(0, n, (nnz0,), generator=g, device=device, dtype=torch.int64)
    c = torch.randint(0, m, (nnz0,), generator=g, device=device, dtype=torch.int64)
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtype]
    v = torch.rand(nnz0, generator=g, device=device, dtype=tdt) + 0.5
    return r, c, v


def _lsr(z, k):
    """Logical right shift of an int64 tensor holding a uint64 bit pattern."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_device(seed, start, count, device):
    """splitmix64(seed, index) for index = start .. start+count-1 as int64 bit patterns (torch has no uint64
    arithmetic; int64 multiplication wraps the same way).  Bit-identical to the numpy splitmix64 above."""
    import torch

    def c(v):   # uint64 constant as the int64 with the same bits
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v
    idx = torch.arange(start + 1, start + 1 + count, dtype=torch.int64, device=device)
    z = idx * c(0x9E3779B97F4A7C15) + c(seed)
    z = (z ^ _lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _umod(z, n):
    """(uint64 bit pattern z) mod n for 0 < n < 2^62, in int64 arithmetic: z = 2 h + b."""
    return ((_lsr(z, 1) % n) * 2 + (z & 1)) % n


def random_uniform_coo_counter_device(n, m, start, count, device, dtype="f64", seed=42, integer_values=False):
    """Triples start .. start+count-1 of random_uniform_coo(n, m, *, dtype, seed), generated on the device
    from the counter PRNG: any rank can produce any slice of the same global list (the C5 leg of bench.py).
    integer_values: values in -9..9 as floats (every sum exact in any order: the parity gate's data);
    otherwise the same uniform [0.5, 1.5) values as the host generator, bit for bit."""
    import torch
    r = _umod(splitmix64_device(seed, start, count, device), n)
    c = _umod(splitmix64_device(seed + 1, start, count, device), m)
    z = splitmix64_device(seed + 2, start, count, device)
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtype]
    if integer_values:
        v = (_umod(z, 19) - 9).to(tdt)
    else:
        v = (_lsr(z, 11).to(torch.float64) * (1.0 / (1 << 53)) + 0.5).to(tdt)
    return r, c, v


def rmat_coo_device(scale, edge_factor, device, dtype="f32", seed=42, probs=(0.57, 0.19, 0.19, 0.05)):
    """C3 shape on the device: RMAT, one quadrant draw per bit level."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cnt = (1 << scale) * edge_factor
    r = torch.zeros(cnt, dtype=torch.int64, device=device)
    c = torch.zeros(cnt, dtype=torch.int64, device=device)
    a, b, cc, _ = probs
    for _ in range(scale):
        u = torch.rand(cnt, generator=g, device=device, dtype=torch.float32)
        r = (r << 1) | (u >= a + b).to(torch.int64)
        c = (c << 1) | (((u >= a) & (u < a + b)) | (u >= a + b + cc)).to(torch.int64)
        del u
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtype]
    v = torch.rand(cnt, generator=g, device=device, dtype=tdt) + 0.5
    return r, c, v


def banded_coo_device(n, half_bw, device, dtype="f64", seed=42):
    """C4 shape on the device: 2*half_bw+1 entries per row, clipped at the edges, row-major."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    i = torch.arange(n, dtype=torch.int64, device=device)
    d = torch.arange(-half_bw, half_bw + 1, dtype=torch.int64, device=device)
    c = (i[:, None] + d[None, :]).reshape(-1)
    r = i[:, None].expand(n, 2 * half_bw + 1).reshape(-1)
    ok = (c >= 0) & (c < n)
    r, c = r[ok].contiguous(), c[ok].contiguous()
    tdt = {"f64": torch.float64, "f32": torch.float32}[dtype]
    v = torch.rand(r.numel(), generator=g, device=device, dtype=tdt) + 0.5
    return r, c, v

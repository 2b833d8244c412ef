"""Row-sharded SpMV across the GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference has no distributed layer (Futhark targets one device); this is the
multi-GPU step BASELINE.json's north_star asks for.  Rows are independent given x,
so the only data-path exchange is x itself:

  * rank k owns a contiguous block of rows and the matching slice of x;
  * at set-up every rank finds which x entries its rows actually reference
    (unique column ids), and the ranks tell each other once which of their
    entries are wanted where (ExchangePlan);
  * per product: each rank packs the entries its peers want, all pairs swap
    them with grouped point-to-point sends (xGMI is a full mesh of
    point-to-point links, so every pair uses its own link concurrently),
    then the ordinary single-GPU kernel runs on the local rows against
    [owned slice | received entries].
  * when the rows reference most of x anyway (random matrices) the plan
    degenerates to "every rank sends its whole slice to everyone", i.e. an
    all-gather, done with one RCCL all_gather_into_tensor when slices are equal.

Local column ids are remapped once (owned: c - lo; remote: n_owned + position in
the sorted ghost list), so the per-step path has no index translation.
"""
import numpy as np


class HostStagedDist:
    """torch.distributed look-alike for REHEARSING the N>1 path with several ranks on one GPU (backend gloo,
    CUDA tensors): every collective and transfer goes through CPU copies.  gloo's own handling of CUDA tensors
    hung intermittently on the GPU box (all_gather, all_gather_into_tensor) with two ranks sharing a device; on
    CPU tensors it is reliable.  Not used with nccl/RCCL, where the tensors stay on the devices."""

    isend, irecv = "isend", "irecv"

    class _Work:
        def __init__(self, works, recvs):
            self.works, self.recvs = works, recvs

        def wait(self):
            for w in self.works:
                w.wait()
            for dst, host in self.recvs:
                dst.copy_(host)
            self.works, self.recvs = [], []

    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp

    @staticmethod
    def P2POp(op, tensor, peer):
        return (op, tensor, peer)

    def get_rank(self):
        return self.d.get_rank()

    def get_world_size(self):
        return self.d.get_world_size()

    def get_backend(self):
        return self.d.get_backend()

    def barrier(self):
        self.d.barrier()

    def all_reduce(self, t, op=None):
        c = t.cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        t.copy_(c)

    def all_gather(self, outs, t):
        co = [o.cpu() for o in outs]
        self.d.all_gather(co, t.cpu())
        for o, c in zip(outs, co):
            o.copy_(c)

    def all_gather_into_tensor(self, out, t):
        c = out.cpu()
        self.d.all_gather_into_tensor(c, t.cpu().contiguous())
        out.copy_(c)

    def batch_isend_irecv(self, ops):
        import torch
        real, recvs = [], []
        for kind, t, peer in ops:
            if kind == "isend":
                real.append(self.d.P2POp(self.d.isend, t.cpu().contiguous(), peer))
            else:
                host = torch.empty(t.shape, dtype=t.dtype)
                recvs.append((t, host))
                real.append(self.d.P2POp(self.d.irecv, host, peer))
        return [self._Work(self.d.batch_isend_irecv(real), recvs)]


def partition_rows_equal_nnz(row_off, parts):
    """Row-block boundaries with (nearly) equal stored entries per part.
    row_off: the reference-convention inclusive prefix sum (length n), host array.
    Returns parts+1 ascending row indices; splitter g = upper_bound(row_off, g*nnz/parts)."""
    row_off = np.asarray(row_off, dtype=np.int64)
    n = len(row_off)
    nnz = int(row_off[-1]) if n else 0
    b = [0]
    for g in range(1, parts):
        b.append(int(np.searchsorted(row_off, (g * nnz) // parts, side="right")))
    b.append(n)
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return b


def coarse_row_splitters(dist, n_glob, rs, parts, bins_log2=16):
    """Row-block boundaries with nearly equal triple counts per part, from a coarse global
    row histogram (<= 2^bins_log2 bins, one all-reduce).  rs: this rank's row ids (any
    device).  Every rank returns the same parts+1 ascending boundaries."""
    import torch
    if parts == 1:
        return [0, n_glob]
    shift = max(0, int(n_glob - 1).bit_length() - bins_log2) if n_glob > 0 else 0
    nb = ((n_glob - 1) >> shift) + 1 if n_glob > 0 else 1
    h = torch.bincount(rs >> shift, minlength=nb).to(torch.int64) if rs.numel() else \
        torch.zeros(nb, dtype=torch.int64, device=rs.device)
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(h)
    cum = torch.cumsum(h, 0)
    total = int(cum[-1].item())
    targets = torch.tensor([(g * total) // parts for g in range(1, parts)], dtype=torch.int64, device=rs.device)
    cut = torch.searchsorted(cum, targets, right=False) + 1             # bins [0, cut) hold >= target triples
    b = [0] + [min(int(c) << shift, n_glob) for c in cut.tolist()] + [n_glob]
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return b


def redistribute_coo(dist, bounds, rs, cs, vs):
    """The one exchange step of distributed assembly: every rank holds an arbitrary chunk of
    the global triple list; afterwards rank k holds exactly the triples whose row lies in
    [bounds[k], bounds[k+1]), ordered by (source rank, position in the source chunk) -- i.e.
    in the order of the concatenated global input, so duplicates are later summed in the same
    order as on one device.  All pairs swap with grouped point-to-point transfers."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    if world == 1:
        return rs, cs, vs
    rank = dist.get_rank()
    dev = rs.device
    bt = torch.tensor(list(bounds), dtype=torch.int64, device=dev)
    dest = torch.searchsorted(bt[1:-1].contiguous(), rs, right=True)    # 0..world-1
    counts = torch.bincount(dest, minlength=world).to(torch.int64)
    allc = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, counts)
    send_counts = [int(v) for v in counts.tolist()]
    recv_counts = [int(allc[p][rank].item()) for p in range(world)]
    # one stable sort of the destinations (source order kept inside each part), then one gather per array
    # instead of a boolean select per (array, destination)
    order = torch.argsort(dest, stable=True)
    offs = [0]
    for k in send_counts:
        offs.append(offs[-1] + k)
    del dest
    outs = []
    for arr in (rs, cs, vs):
        packed = arr.index_select(0, order)
        parts = [packed[offs[p]:offs[p + 1]] for p in range(world)]
        recv = [parts[p] if p == rank else torch.empty(recv_counts[p], dtype=arr.dtype, device=dev)
                for p in range(world)]
        ops = []
        for p in range(world):
            if p == rank:
                continue
            if send_counts[p]:
                ops.append(dist.P2POp(dist.isend, parts[p], p))
            if recv_counts[p]:
                ops.append(dist.P2POp(dist.irecv, recv[p], p))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        outs.append(torch.cat(recv))
        del parts, recv, packed
    return tuple(outs)


class ExchangePlan:
    """Who sends which x entries to whom.  Pure torch + torch.distributed; works on
    CPU tensors with gloo (tests) and CUDA tensors with nccl/RCCL (product)."""

    ALLGATHER_FRACTION = 0.5

    def __init__(self, dist, bounds, needed_cols, device, force_mode=None):
        """bounds: world+1 ascending global indices; rank k owns x[bounds[k]:bounds[k+1]].
        needed_cols: sorted unique global column ids referenced by this rank's rows."""
        import torch
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.bounds = [int(b) for b in bounds]
        self.device = device
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.n_own = self.hi - self.lo
        m = self.bounds[-1]
        remote = needed_cols[(needed_cols < self.lo) | (needed_cols >= self.hi)]
        n_remote_total = m - self.n_own
        # one collective decision so that every rank takes the same path
        frac = torch.tensor([float(remote.numel()) / max(n_remote_total, 1)], dtype=torch.float64, device=device)
        if dist is not None and self.world > 1:
            dist.all_reduce(frac, op=dist.ReduceOp.MAX)
        mode = "allgather" if float(frac.item()) >= self.ALLGATHER_FRACTION else "halo"
        if force_mode:
            mode = force_mode
        if self.world == 1:
            mode = "none"
        self.mode = mode
        bt = torch.tensor(self.bounds, dtype=torch.int64, device=device)
        if mode in ("allgather", "none"):
            # x_ext is the whole vector in global numbering
            self.x_len = m
            self.own_offset = self.lo
            self.ghost = None
            self.equal_slices = len({self.bounds[k + 1] - self.bounds[k] for k in range(self.world)}) == 1
            self.send_idx = None
            self.send_counts = [0 if k == self.rank else self.n_own for k in range(self.world)]
            self.recv_counts = [0 if k == self.rank else self.bounds[k + 1] - self.bounds[k] for k in range(self.world)]
            return
        # ---- halo mode
        self.ghost = remote.contiguous()
        self.x_len = self.n_own + int(self.ghost.numel())
        self.own_offset = 0
        owner = torch.searchsorted(bt, self.ghost, right=True) - 1          # owning rank of each ghost
        recv_counts = torch.bincount(owner, minlength=self.world).to(torch.int64)
        self.recv_counts = [int(v) for v in recv_counts.tolist()]
        # tell every peer how many (then which) of its entries are wanted here
        all_counts = [torch.zeros(self.world, dtype=torch.int64, device=device) for _ in range(self.world)]
        dist.all_gather(all_counts, recv_counts)
        self.send_counts = [int(all_counts[p][self.rank].item()) for p in range(self.world)]
        want_from = torch.split(self.ghost, self.recv_counts)               # ghost is sorted => grouped by owner
        send_lists = [torch.empty(self.send_counts[p], dtype=torch.int64, device=device) for p in range(self.world)]
        ops = []
        for p in range(self.world):
            if p == self.rank:
                continue
            if self.recv_counts[p]:
                ops.append(dist.P2POp(dist.isend, want_from[p].contiguous(), p))
            if self.send_counts[p]:
                ops.append(dist.P2POp(dist.irecv, send_lists[p], p))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.send_idx = (torch.cat(send_lists) - self.lo) if sum(self.send_counts) else \
            torch.empty(0, dtype=torch.int64, device=device)
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.n_own)
        self._sendbuf = None
        self._ops_key = None

    # -- column numbering of the local matrix
    def remap_cols(self, cols):
        torch = self.torch
        if self.mode != "halo":
            return cols
        own = (cols >= self.lo) & (cols < self.hi)
        gpos = torch.searchsorted(self.ghost, cols)
        return torch.where(own, cols - self.lo, gpos + self.n_own)

    def new_x(self, dtype):
        """The extended vector [.. owned slice ..] this rank multiplies against, and
        the view of it that holds the owned slice."""
        x_ext = self.torch.zeros(self.x_len, dtype=dtype, device=self.device)
        return x_ext, x_ext[self.own_offset:self.own_offset + self.n_own]

    def exchange(self, x_ext):
        """Fill the non-owned part of x_ext from the peers (in place)."""
        torch, dist = self.torch, self.dist
        if self.mode == "none":
            return x_ext
        own = x_ext[self.own_offset:self.own_offset + self.n_own]
        if self.mode == "allgather":
            if self.equal_slices and getattr(self, "allgather_impl", "collective") == "collective":
                # the collective's input may not alias its output: a staging buffer made once per x_ext (no
                # allocation per product), filled by a device copy on the same stream
                key = (x_ext.data_ptr(), x_ext.dtype, tuple(x_ext.shape))
                if getattr(self, "_stage_key", None) != key:
                    self._stage, self._stage_key = torch.empty_like(own), key
                self._stage.copy_(own)
                dist.all_gather_into_tensor(x_ext, self._stage)
                return x_ext
            # all pairs at once: every rank sends its slice to every other rank directly (xGMI is a mesh of
            # point-to-point links; a ring all-gather is bound by one link).  The transfer list is made once.
            key = (x_ext.data_ptr(), x_ext.dtype, tuple(x_ext.shape))
            if getattr(self, "_ag_key", None) != key:
                ops = []
                for p in range(self.world):
                    if p == self.rank:
                        continue
                    if self.n_own:
                        ops.append(dist.P2POp(dist.isend, own, p))
                    if self.bounds[p + 1] > self.bounds[p]:
                        ops.append(dist.P2POp(dist.irecv, x_ext[self.bounds[p]:self.bounds[p + 1]], p))
                self._ag_ops, self._ag_key = ops, key
            if self._ag_ops:
                for w in dist.batch_isend_irecv(self._ag_ops):
                    w.wait()
            return x_ext
        for w in self.exchange_start(x_ext):
            w.wait()
        return x_ext

    def exchange_start(self, x_ext):
        """halo mode: pack and post the point-to-point transfers; returns the work handles.
        The pack buffer and the list of transfers are made at the first call for a given x_ext and
        reused (the views they hold stay valid: x_ext is persistent)."""
        torch, dist = self.torch, self.dist
        key = (x_ext.data_ptr(), x_ext.dtype, tuple(x_ext.shape))
        if getattr(self, "_ops_key", None) != key:
            own = x_ext[:self.n_own]
            self._own = own
            self._sendbuf = torch.empty((int(self.send_idx.numel()),) + tuple(x_ext.shape[1:]), dtype=x_ext.dtype,
                                        device=x_ext.device)
            recv = x_ext[self.n_own:]
            ops, so, ro = [], 0, 0
            for p in range(self.world):
                sc, rc = self.send_counts[p], self.recv_counts[p]
                if p != self.rank:
                    if sc:
                        ops.append(dist.P2POp(dist.isend, self._sendbuf[so:so + sc], p))
                    if rc:
                        ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc], p))
                so += sc
                ro += rc
            self._ops, self._ops_key = ops, key
        if self._sendbuf.numel():
            torch.index_select(self._own, 0, self.send_idx, out=self._sendbuf)
        return dist.batch_isend_irecv(self._ops) if self._ops else []


class DistCsr:
    """This rank's row block as ordinary single-GPU CSR records plus the plan that keeps
    its x up to date.

    halo mode: the block is split by column ownership, A = [A_loc | A_gh].  A_loc (columns
    this rank owns) is multiplied while the halo entries are still on the wire; A_gh holds
    only the rows that touch a remote column (compact row numbering) and is added afterwards:
        y = A_loc x_own ;  y[rows_g] += A_gh x_ghost
    all-gather / single-rank mode: one record over the whole x."""

    def __init__(self, M, plan, A, n_loc, torch_dtype, A_gh=None, rows_g=None):
        self.M, self.plan, self.A, self.n_loc = M, plan, A, n_loc
        self.A_gh, self.rows_g = A_gh, rows_g
        # One stream for everything a product does.  The library runs on its context's stream, torch ops and
        # the RCCL transfers (whose wait() orders torch's current stream only) on torch's: if they differed the
        # product on received entries could start before they have landed.  So the context adopts torch's
        # current stream here (it first drains its own); nothing is left to the caller.
        import torch as _torch
        if _torch.cuda.is_available() and str(plan.device).startswith("cuda"):
            M.ctx.set_stream(_torch.cuda.current_stream(_torch.device(plan.device)).cuda_stream)
        self.mode = plan.mode
        self.x_len = plan.x_len if A_gh is None else plan.n_own
        self.x_ext, self.x_own = plan.new_x(torch_dtype)
        self.x_ghost = self.x_ext[plan.n_own:] if A_gh is not None else None
        self.nnz = A.stored_dims()[2] + (A_gh.stored_dims()[2] if A_gh is not None else 0)
        # per-step state made once: the result vector and the library's handles on the torch storage
        # (one C call per product, no array created in the loop)
        import torch
        ctx = M.ctx
        self.y = torch.zeros(n_loc, dtype=torch_dtype, device=self.x_ext.device)
        self._y_h = ctx.array(self.y, M.t)
        self._x_h = ctx.array(self.x_own if A_gh is not None else self.x_ext, M.t)
        self._xg_h = ctx.array(self.x_ghost, M.t) if A_gh is not None and self.x_ghost.numel() else None
        self._rows_h = ctx.array(self.rows_g, "i64") if rows_g is not None and rows_g.numel() else None

    @classmethod
    def from_local_coo(cls, M, dist, n_glob, m_glob, row_lo, row_hi, rs, cs, vs, force_mode=None, split=True):
        """rs/cs/vs: this rank's COO triples (torch CUDA tensors, GLOBAL indices, rows in
        [row_lo,row_hi)).  Square matrices: x is partitioned like the rows.  Rectangular ones (n_glob !=
        m_glob): x is cut into world equal slices (the last takes the remainder)."""
        import torch
        dev = rs.device
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        if n_glob == m_glob:
            mine = torch.tensor([row_hi], dtype=torch.int64, device=dev)
            his = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            if world > 1:
                dist.all_gather(his, mine)
            else:
                his = [mine]
            bounds = [0] + [int(h.item()) for h in his]
            assert bounds[-1] == m_glob and bounds[rank] == row_lo
        else:
            per = -(-m_glob // world)
            bounds = [min(k * per, m_glob) for k in range(world)] + [m_glob]
        # the columns this rank's rows reference, ascending.  A single rank exchanges nothing; otherwise a byte
        # mask over the columns (one scatter, one compaction) is cheaper than sorting the ids when it fits
        if world == 1:
            needed = cs[:0]
        elif m_glob <= (1 << 31):
            mask = torch.zeros(m_glob, dtype=torch.bool, device=dev)
            mask[cs] = True
            needed = mask.nonzero().flatten()
            del mask
        else:
            needed = torch.unique(cs)
        plan = ExchangePlan(dist, bounds, needed, dev, force_mode)
        del needed
        n_loc = row_hi - row_lo
        if plan.mode != "halo" or not split:
            A = M.sr.sparse(n_loc, plan.x_len, (rs - row_lo if row_lo else rs, plan.remap_cols(cs), vs))
            return cls(M, plan, A, n_loc, vs.dtype)
        own = (cs >= plan.lo) & (cs < plan.hi)
        A = M.sr.sparse(n_loc, plan.n_own, ((rs[own] - row_lo).contiguous(), (cs[own] - plan.lo).contiguous(),
                                            vs[own].contiguous()))
        gh = ~own
        rows_g = torch.unique(rs[gh])                                   # sorted global rows touching a ghost
        n_gh = max(int(plan.ghost.numel()), 1)
        A_gh = M.sr.sparse(int(rows_g.numel()), n_gh,
                           (torch.searchsorted(rows_g, rs[gh]).contiguous(),
                            torch.searchsorted(plan.ghost, cs[gh]).contiguous(), vs[gh].contiguous()))
        return cls(M, plan, A, n_loc, vs.dtype, A_gh, (rows_g - row_lo).contiguous())

    @classmethod
    def from_coo_chunks(cls, M, dist, n_glob, m_glob, rs, cs, vs, bounds=None, force_mode=None, split=True):
        """Distributed assembly: every rank passes an arbitrary chunk of the global COO list
        (GLOBAL indices).  Splitters come from a coarse global row histogram unless given;
        one all-to-all-v of triples; then the ordinary single-GPU assembly on each rank."""
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        if bounds is None:
            bounds = coarse_row_splitters(dist, n_glob, rs, world)
        rs, cs, vs = redistribute_coo(dist, bounds, rs, cs, vs)
        D = cls.from_local_coo(M, dist, n_glob, m_glob, bounds[rank], bounds[rank + 1], rs, cs, vs, force_mode, split)
        D.bounds = list(bounds)
        return D

    def prepare(self, expected_products):
        """How many products will follow (sparse_amd_csr_*_prepare on this rank's records): a solver says a large
        number and the records build their faster SpMV forms now instead of at their 16th product."""
        self.M.sr.prepare(self.A, expected_products)
        if self.A_gh is not None:
            self.M.sr.prepare(self.A_gh, expected_products)
        return self

    def nnz_offset(self):
        """Stored entries on lower ranks (one all-gather of a scalar per rank)."""
        import torch
        dist = self.plan.dist
        if dist is None or self.plan.world == 1:
            return 0
        dev = self.x_ext.device
        mine = torch.tensor([self.nnz], dtype=torch.int64, device=dev)
        allv = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.plan.world)]
        dist.all_gather(allv, mine)
        return sum(int(v.item()) for v in allv[:self.plan.rank])

    def exchange(self, x_own=None):
        if x_own is not None and x_own.data_ptr() != self.x_own.data_ptr():
            self.x_own.copy_(x_own)
        return self.plan.exchange(self.x_ext)

    def tune_exchange(self, reps=5):
        """All-gather mode with equal slices can move x by one RCCL collective or by grouped point-to-point
        transfers (all pairs at once).  Which is faster depends on the algorithm RCCL picks for the size
        (a ring is bound by one xGMI link, the all-pairs form uses all seven), so both are timed here, a few
        exchanges each, and every rank adopts the same winner (max over ranks).  Returns {impl: seconds}."""
        import time
        import torch
        plan, dist = self.plan, self.plan.dist
        if plan.mode != "allgather" or not plan.equal_slices or dist is None:
            return {}
        on_gpu = self.x_ext.is_cuda
        res = {}
        for impl in ("collective", "p2p"):
            plan.allgather_impl = impl
            plan.exchange(self.x_ext)                       # warm-up (connections, transfer list)
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.exchange(self.x_ext)
            if on_gpu:
                torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.x_ext.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[impl] = float(t.item()) / reps
        plan.allgather_impl = min(res, key=res.get)
        return res

    def local_smvm(self, x_ext=None):
        """The dominant kernel alone: this rank's rows against data already in place.  Returns the
        persistent result vector DistCsr.y (torch tensor, overwritten by the next product)."""
        if x_ext is not None and x_ext.data_ptr() != self.x_ext.data_ptr():
            self.x_ext.copy_(x_ext)
        self.M.sr.smvm_into(self._y_h, self.A, self._x_h)
        return self.y

    def smvm(self, x_own=None):
        """y[rows of this rank] = A[rows,:] x.  x_own: this rank's slice of x (ideally the
        view DistCsr.x_own itself, then nothing is copied).  Returns DistCsr.y (overwritten by the
        next product)."""
        if self.A_gh is None:
            self.exchange(x_own)
            return self.local_smvm()
        if x_own is not None and x_own.data_ptr() != self.x_own.data_ptr():
            self.x_own.copy_(x_own)
        works = self.plan.exchange_start(self.x_ext)       # halo entries go on the wire ...
        self.M.sr.smvm_into(self._y_h, self.A, self._x_h)  # ... while the owned columns are multiplied
        for w in works:
            w.wait()
        if self._rows_h is not None:
            self.M.sr.smvm_into(self._y_h, self.A_gh, self._xg_h, self._rows_h)
        return self.y

    def smm(self, B_own):
        """C[rows of this rank][k] = A[rows,:] B for a dense B whose rows are partitioned
        like x; B_own: this rank's [n_own][k] block (row-major).  Same exchange plan as smvm,
        moving rows of B instead of scalars."""
        import torch
        plan, dev = self.plan, self.x_ext.device
        k = int(B_own.shape[1])
        B_ext = torch.empty((plan.x_len, k), dtype=B_own.dtype, device=dev)
        B_ext[plan.own_offset:plan.own_offset + plan.n_own].copy_(B_own)
        if self.A_gh is None:
            plan.exchange(B_ext)
            return self.M.sr.smm(self.A, B_ext)
        works = plan.exchange_start(B_ext)
        C = self.M.sr.smm(self.A, B_ext[:plan.n_own])
        for w in works:
            w.wait()
        if self.rows_g.numel():
            Cg = self.M.sr.smm(self.A_gh, B_ext[plan.n_own:])
            torch.as_tensor(C, device=dev).index_add_(0, self.rows_g, torch.as_tensor(Cg, device=dev))
        return C


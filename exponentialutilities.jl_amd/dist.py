"""Multi-GPU plumbing for batches of independent problems (BASELINE config 5).

The Krylov path shards only across independent (A, b) problems: one process per GPU, every rank
runs its own contiguous block of problems through the HIP path, and the ONLY collective is the final
gather of the result block (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
torch.distributed is plumbing here -- nothing in this file computes.

One exception, for BASELINE config 3 at its literal size (SURVEY.md §8d "C3 fit"): a dense operator that does not fit one
GPU is ROW-SHARDED (RowShardedDense): the only multi-GPU exchange inside a problem is one all-gather of the n-vector per
operator application; everything else of the Krylov iteration runs replicated and identically on every rank.
"""
import time


def shard_range(nprob, world_size, rank):
    """Contiguous, balanced block of problem indices owned by `rank` (first `nprob % world` ranks get one more)."""
    base, extra = divmod(int(nprob), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(nprob, world_size):
    return [shard_range(nprob, world_size, r)[1] - shard_range(nprob, world_size, r)[0] for r in range(world_size)]


def gather_columns(local_block, nprob, group=None):
    """All ranks contribute their (n x nlocal) result block; every rank gets the (n x nprob) matrix, columns in
    problem order.  One all_gather of fixed-size padded blocks (a single large collective, not one per problem)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_block
    world = dist.get_world_size(group)
    sizes = shard_sizes(nprob, world)
    n = local_block.shape[0]
    width = max(sizes)
    # work on the (nlocal x n) transpose so that each problem's result is one contiguous row
    send = torch.zeros((width, n), dtype=local_block.dtype, device=local_block.device)
    send[: local_block.shape[1]].copy_(local_block.t())
    recv = torch.empty((world * width, n), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    rows = [recv[r * width: r * width + sizes[r]] for r in range(world)]
    return torch.cat(rows, dim=0).t()


def aggregate_throughput(units_local, elapsed_local, device=None, group=None):
    """Whole-job rate: SUM of the units over ranks / MAX of the elapsed time over ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(units_local), float(elapsed_local)
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(u.item()), float(t.item())


def run_sharded(nprob, solve_one, make_block, group=None, rank=None, world_size=None):
    """Run problems [lo, hi) of this rank through `solve_one(i) -> vector` and gather all results.
    `make_block(list_of_vectors) -> (n x nlocal) tensor`.  Returns (results n x nprob, units, seconds)."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(nprob, world_size, rank)
    t0 = time.perf_counter()
    cols = [solve_one(i) for i in range(lo, hi)]
    elapsed = time.perf_counter() - t0
    block = make_block(cols)
    return gather_columns(block, nprob, group), hi - lo, elapsed


class RowShardedDense:
    """y = A x for a dense n x n operator whose ROWS are spread over the ranks of `group` (config 3: n = 2e5 fp64 is 320 GB,
    more than the 288 GB of one MI355X).  Rank r holds rows [lo_r, hi_r) of A (shard_range(n, world, r)) as a tensor `rows`
    of shape (hi_r - lo_r, n).  On the GPU the block must be COLUMN-MAJOR (``torch.empty(n, nloc).t()``: the library's layout,
    krylov_phiv_adaptive.jl works on Julia matrices) and an application is the library's own dense GEMV on it
    (expv_mi_gemv_block: the kernel of the dense operator's mul!, no vendor BLAS) followed by ONE all-gather of the result
    pieces (n * 8 B = 1.6 MB at n = 2e5; RCCL over xGMI with backend "nccl").  Every rank then holds the same y, so a Krylov
    iteration driven by this operator runs replicated: the same kernels on the same data on every rank (deterministic), no
    other communication, and every rank ends with the full result.  `operator(eu, ctx)` wraps it as the library's matrix-free
    operator (expv_mi_op_create_callback).  A CPU tensor (the gloo tests of the collective plumbing) takes torch.mv."""

    def __init__(self, rows, n, group=None, stage_through_host=False, collective_at_world_1=False):
        import torch
        import torch.distributed as dist
        self.rows = rows
        self.n = int(n)
        self.group = group
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.lo, self.hi = shard_range(self.n, self.world, self.rank)
        if tuple(rows.shape) != (self.hi - self.lo, self.n):
            raise ValueError("rank %d holds rows [%d, %d): expected a (%d, %d) block, got %s"
                             % (self.rank, self.lo, self.hi, self.hi - self.lo, self.n, tuple(rows.shape)))
        self.on_gpu = bool(rows.is_cuda)
        if self.on_gpu and self.hi > self.lo and not (rows.stride(0) == 1 and rows.stride(1) >= self.hi - self.lo):
            raise ValueError("a device-resident row block must be column-major (torch.empty(n, nloc).t()): strides %s"
                             % (tuple(rows.stride()),))
        self.width = max(shard_sizes(self.n, self.world))          # equal-size pieces for ONE fixed-size collective
        self.stage = bool(stage_through_host)                      # gloo has no device all-gather: CPU tests / fallback
        self._send = torch.zeros(self.width, dtype=rows.dtype, device=rows.device)
        self._recv = torch.empty(self.world * self.width, dtype=rows.dtype, device=rows.device)
        self.applications = 0
        self.collectives = 0
        # a single rank needs no exchange; with this flag it runs the all-gather anyway (a one-rank RCCL group on a one-GPU box
        # executes the same collective code as N ranks do)
        self.always_collective = bool(collective_at_world_1) and self.dist is not None
        self._ctx = None
        self._lib = None
        if self.on_gpu:
            # column splits like expv_mi_op_create_dense picks them: >= 1024 workgroups for a block with few row tiles
            nloc = self.hi - self.lo
            rows_per_block = 256 * (16 // rows.element_size())     # a lane moves one 16-byte pack: 1 / 2 / 4 rows (capi.hip: op_create_dense)
            gx = max(1, -(-nloc // rows_per_block))
            self._nsplit = 1 if nloc < 64 else min(64, max(1, -(-1024 // gx)))
            self._scratch = torch.empty(max(1, self._nsplit * nloc), dtype=rows.dtype, device=rows.device) if self._nsplit > 1 else None

    @staticmethod
    def column_major(rows):
        """a (nloc, n) tensor laid out column-major (one transposing copy; build large blocks in this layout directly)"""
        return rows if rows.stride(0) == 1 else rows.t().contiguous().t()

    def _local_gemv(self, x):
        """rows [lo, hi) of A x into the head of the send buffer."""
        import torch
        nloc = self.hi - self.lo
        if not self.on_gpu:
            yl = torch.mv(self.rows, x)                            # CPU tensors: test plumbing only
            self._send[:nloc].copy_(yl)
            return
        if self._ctx is None:
            raise RuntimeError("RowShardedDense: call operator(eu, ctx) first (the local GEMV runs on the library's stream)")
        code = _DTYPE_CODE[self.rows.dtype]                        # EXPV_MI_F64 / C64 / F32 / C32: the block's own element type
        if x.dtype != self.rows.dtype:
            raise TypeError("RowShardedDense: x is %s, the row block %s" % (x.dtype, self.rows.dtype))
        if not x.is_contiguous():
            x = x.contiguous()
        rc = self._lib.expv_mi_gemv_block(self._ctx._h, code, nloc, self.n, self.rows.data_ptr(), self.rows.stride(1) if self.n > 1 else max(nloc, 1),
                                          x.data_ptr(), self._send.data_ptr(),
                                          self._scratch.data_ptr() if self._scratch is not None else None, self._nsplit)
        if rc != 0:
            msg = self._lib.expv_mi_last_error(self._ctx._h)
            raise RuntimeError("expv_mi_gemv_block failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def matvec(self, x):
        import torch
        self.applications += 1
        self._local_gemv(x)
        if self.world == 1 and not self.always_collective:
            return self._send[: self.hi - self.lo]
        self.collectives += 1
        if self.stage:
            send, recv = self._send.cpu(), torch.empty(self.world * self.width, dtype=self._send.dtype)
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            self._recv.copy_(recv)
        else:
            self.dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        if self.n == self.world * self.width:
            return self._recv
        sizes = shard_sizes(self.n, self.world)
        return torch.cat([self._recv[r * self.width: r * self.width + sizes[r]] for r in range(self.world)])

    def operator(self, eu, ctx=None, ishermitian=False):
        """The library operator (device vectors in, device vectors out, on the library's stream)."""
        if self.on_gpu:
            import torch
            self._ctx = ctx or eu.default_context()
            self._lib = eu._lib.load()
            # the row block was produced on torch's stream (a copy, a transposition, a generator); the local GEMV reads it on the
            # library's own non-blocking stream, which does not wait for the legacy default stream: order them once, here
            torch.cuda.synchronize(self.rows.device)
        return eu.MIOperator(None, ctx, dtype=_np_dtype(self.rows.dtype), ishermitian=ishermitian, matvec=self.matvec,
                             shape=(self.n, self.n))


class _DtypeCodes(dict):
    """torch dtype -> the library's element-type code (include/expv_mi.h: EXPV_MI_F64 = 0, C64 = 1, F32 = 2, C32 = 3)"""

    def __missing__(self, key):
        import torch
        self.update({torch.float64: 0, torch.complex128: 1, torch.float32: 2, torch.complex64: 3})
        if key not in self:
            raise TypeError("RowShardedDense: unsupported element type %s" % (key,))
        return self[key]


_DTYPE_CODE = _DtypeCodes()


def _np_dtype(torch_dtype):
    import numpy as np
    import torch
    return {torch.float64: np.float64, torch.complex128: np.complex128, torch.float32: np.float32,
            torch.complex64: np.complex64}[torch_dtype]

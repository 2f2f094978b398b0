"""Multi-GPU plumbing for batches of independent problems (BASELINE config 5).

The Krylov path shards only across independent (A, b) problems: one process per GPU, every rank
runs its own contiguous block of problems through the HIP path, and the ONLY collective is the final
gather of the result block (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
torch.distributed is plumbing here -- nothing in this file computes.
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

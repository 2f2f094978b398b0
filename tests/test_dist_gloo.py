"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: problem sharding, the single final gather
and the whole-job throughput aggregation used by bench.py at N > 1.  The per-problem solver here is the
oracle (the checker); the HIP path is exercised by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import expv_mi_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nprob, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi_dist", os.path.join(ROOT, "exponentialutilities.jl_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    from oracle import krylov_oracle as ko
    from tests._util import c2_operator
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def solve_one(i):
            rng = np.random.default_rng(7000 + i)
            A = c2_operator(n) * (1 + 0.1 * rng.random())
            b = rng.standard_normal(n)
            return ko.expv(1.0, A.tocsr(), b, m=12, ishermitian=False)

        make_block = lambda cols: torch.as_tensor(np.stack(cols, axis=1) if cols else np.zeros((n, 0)))
        res, units, elapsed = D.run_sharded(nprob, solve_one, make_block)
        tot_units, max_t = D.aggregate_throughput(units, elapsed)
        np.save(os.path.join(out_dir, f"res_{rank}.npy"), res.numpy())
        np.save(os.path.join(out_dir, f"agg_{rank}.npy"), np.array([tot_units, max_t, units, elapsed]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nprob", [5, 8])
def test_sharded_batch_two_ranks(tmp_path, nprob):
    world, n = 2, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nprob, n, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "res_0.npy")
    r1 = np.load(tmp_path / "res_1.npy")
    assert r0.shape == (n, nprob)
    np.testing.assert_array_equal(r0, r1)                      # every rank holds the gathered result
    from oracle import krylov_oracle as ko
    from tests._util import c2_operator
    for i in range(nprob):                                      # columns are in problem order
        rng = np.random.default_rng(7000 + i)
        A = c2_operator(n) * (1 + 0.1 * rng.random())
        b = rng.standard_normal(n)
        np.testing.assert_allclose(r0[:, i], ko.expv(1.0, A.tocsr(), b, m=12, ishermitian=False), rtol=1e-13, atol=1e-15)
    a0, a1 = np.load(tmp_path / "agg_0.npy"), np.load(tmp_path / "agg_1.npy")
    assert a0[0] == a1[0] == nprob                              # SUM of units
    assert a0[1] == a1[1] == max(a0[3], a1[3])                  # MAX of elapsed


def test_shard_range_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi_dist", os.path.join(ROOT, "exponentialutilities.jl_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for nprob in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = D.shard_range(nprob, world, r)
                covered += list(range(lo, hi))
            assert covered == list(range(nprob))
            assert max(D.shard_sizes(nprob, world)) - min(D.shard_sizes(nprob, world)) <= 1

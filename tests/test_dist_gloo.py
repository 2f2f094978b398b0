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


# ---- bench.py's config-5 driver (run_c5) end to end under gloo: sharding, batch call, gather, whole-job aggregation,
#      ranks_seen, per-rank times and the recompute-two-columns verification.  The per-problem solver is a stand-in with
#      the product's signatures (expv_batch / expv) backed by the oracle: the CODE PATH of bench.py is what is tested here,
#      the HIP path behind the same signatures is covered by the -m gpu tests.
class _StubEU:
    @staticmethod
    def expv(t, A, b, m=30, ishermitian=False, **kw):
        from oracle import krylov_oracle as ko
        return ko.expv(t, A.tocsr(), np.asarray(b), m=m, ishermitian=ishermitian)

    @staticmethod
    def expv_batch(t, A0, vals, B, m=30, ctx=None, **kw):
        from oracle import krylov_oracle as ko
        vals = vals.numpy() if hasattr(vals, "numpy") else np.asarray(vals)
        Bn = B.numpy() if hasattr(B, "numpy") else np.asarray(B)
        cols = []
        for p in range(vals.shape[0]):
            Ap = A0.copy()
            Ap.data = vals[p].copy()
            cols.append(ko.expv(t, Ap, Bn[:, p], m=m, ishermitian=False))
        return torch.as_tensor(np.stack(cols, axis=1) if cols else np.zeros((A0.shape[0], 0)))


def _c5_worker(rank, world, port, nprob, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    import json
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = bench.Env(torch, dist, world, rank, "cpu", None)
        args = argparse.Namespace(nprob=nprob, steps=2, warmup=1)
        out = bench.run_c5(args, _StubEU, env, n=n, m=10, do_emit=False)
        json.dump(out, open(os.path.join(out_dir, f"c5_{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nprob", [5, 6])
def test_bench_c5_driver_two_ranks(tmp_path, nprob):
    import json
    world, n = 2, 96
    port = _free_port()
    mp.spawn(_c5_worker, args=(world, port, nprob, n, str(tmp_path)), nprocs=world, join=True)
    o0 = json.load(open(tmp_path / "c5_0.json"))
    o1 = json.load(open(tmp_path / "c5_1.json"))
    assert o0["ranks_seen"] == o1["ranks_seen"] == world
    assert o0["n_gpus"] == world and o0["scaling"] == "strong"
    assert len(o0["per_rank_ms_per_step"]) == world and all(v > 0 for v in o0["per_rank_ms_per_step"])
    assert o0["value"] == o1["value"] > 0                           # SUM of units / MAX of time: the same on every rank
    np.testing.assert_allclose(o0["value"], nprob * 10 * 2 / (o0["ms_per_step"] * 2e-3), rtol=1e-9)
    assert o0["verified"]["max_rel_err"] <= 1e-12 and o0["verified"]["columns_per_rank"] == 2
    assert o0["config"]["nprob"] == nprob


def test_bench_byte_contracts_match_survey():
    """SURVEY.md §8d figures, to the digit: C2 A_B = 64.0 MB, 6.384 GB per expv, 212.8 MB per matvec; per-step mean
    A_B + 8n(j+2) = 204 MB; Lanczos variant 3.14 GB; C4 184 MB per Krylov step."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    n, nnz, m = 1_000_000, 5 * 1_000_000 - 6, 30
    assert abs(bench.a_bytes(n, nnz) - 64.0e6) < 0.1e6
    assert abs(bench.alg_bytes_expv(n, nnz, m) - 6.384e9) < 0.001e9
    assert abs(bench.alg_bytes_expv(n, nnz, m) / m - 212.8e6) < 0.1e6
    assert abs(bench.alg_bytes_step(n, nnz, m) - 204.0e6) < 0.1e6
    assert abs(bench.alg_bytes_expv_window(n, nnz, m, 2) - 3.14e9) < 0.01e9
    assert abs(bench.alg_bytes_kiops(n, nnz, 1, 0, 0) - 184.0e6) < 0.1e6
    sym = bench.c2_operator(50, sym=True)
    assert (sym != sym.T).nnz == 0 and (bench.c2_operator(50) != bench.c2_operator(50).T).nnz > 0


# ---- BASELINE config 3 at its literal size: dense operator row-sharded over the ranks (dist.RowShardedDense) -----------------
def _c3_inputs(n, K=4):
    A = -2.0 * np.eye(n) + np.random.default_rng(4).standard_normal((n, n)) / np.sqrt(n)
    B = np.asfortranarray(np.random.default_rng(5).standard_normal((n, K + 1)))
    return A, B


class _ShardedAsMatrixFree:
    """The operator contract of docs/src/interfaces.md:7-36 (eltype / size / mul! / ishermitian) on top of the sharded matvec."""

    def __init__(self, sh):
        self.sh, self.shape, self.dtype, self.ishermitian = sh, (sh.n, sh.n), np.dtype(np.float64), False

    def __matmul__(self, x):
        return self.sh.matvec(torch.as_tensor(np.ascontiguousarray(x))).numpy().copy()


def _worker_rows(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi_dist", os.path.join(ROOT, "exponentialutilities.jl_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    from oracle import krylov_oracle as ko
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, B = _c3_inputs(n)
        lo, hi = D.shard_range(n, world, rank)
        sh = D.RowShardedDense(torch.as_tensor(np.ascontiguousarray(A[lo:hi])), n)      # this rank never touches the other rows
        x = np.random.default_rng(11).standard_normal(n)
        y = sh.matvec(torch.as_tensor(x)).numpy().copy()
        st = {}
        U = ko.phiv_timestep(np.array([0.5, 1.0]), _ShardedAsMatrixFree(sh), B, adaptive=True, tol=1e-7, stats=st)
        np.save(os.path.join(out_dir, f"y_{rank}.npy"), y)
        np.save(os.path.join(out_dir, f"U_{rank}.npy"), U)
        np.save(os.path.join(out_dir, f"st_{rank}.npy"), np.array([st["num_timesteps"], st["matvecs"], st["m"], sh.applications]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64])
def test_row_sharded_dense_operator_two_ranks(tmp_path, n):
    """Config 3 row-sharded over 2 ranks (ragged and even split): the sharded matvec equals A x on every rank, and the adaptive
    phiv_timestep driven by it (the oracle's controller here; the device engine on the GPU) takes the same steps and gives the
    same snapshots on both ranks as the unsharded operator."""
    world = 2
    mp.spawn(_worker_rows, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    from oracle import krylov_oracle as ko
    A, B = _c3_inputs(n)
    x = np.random.default_rng(11).standard_normal(n)
    y0, y1 = np.load(tmp_path / "y_0.npy"), np.load(tmp_path / "y_1.npy")
    np.testing.assert_array_equal(y0, y1)
    np.testing.assert_allclose(y0, A @ x, rtol=0, atol=1e-13 * np.abs(A @ x).max())
    so = {}
    Uo = ko.phiv_timestep(np.array([0.5, 1.0]), A, B, adaptive=True, tol=1e-7, stats=so)
    U0, U1 = np.load(tmp_path / "U_0.npy"), np.load(tmp_path / "U_1.npy")
    np.testing.assert_array_equal(U0, U1)                      # replicated iteration: bitwise the same on every rank
    s0, s1 = np.load(tmp_path / "st_0.npy"), np.load(tmp_path / "st_1.npy")
    np.testing.assert_array_equal(s0, s1)
    assert tuple(s0[:3]) == (so["num_timesteps"], so["matvecs"], so["m"])
    assert s0[3] >= so["matvecs"]                              # every operator application went through the collective
    assert np.linalg.norm(U0 - Uo) <= 1e-12 * np.linalg.norm(Uo)


# ---- `python bench.py --gpus N` with NO launcher around it must start N ranks itself.  Run as a subprocess
#      with the CPU stand-in solver (tests/standin_eu.py): the launch / rendezvous / sharding / collective / verification code of
#      bench.py is the product's; only the per-problem solver is replaced.
def _run_bench(extra, env_extra=None):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["EXPV_MI_BENCH_STANDIN_OK"] = "tests-only"               # bench.py refuses --standin without it
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--standin",
                        "tests.standin_eu"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), lines


@pytest.mark.parametrize("cfg", [["--config", "c2", "--n", "300"], ["--config", "c5", "--n5", "96", "--nprob", "5"],
                                 ["--config", "c3", "--n3", "101"]])
def test_bench_gpus_2_launches_two_ranks_by_itself(cfg):
    r, out, lines = _run_bench(["--gpus", "2"] + cfg)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1                                       # rank 0 prints ONE JSON line
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert len(out["devices"]) == 2 and len(set(out["devices"])) == 2
    assert len(out["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in out["per_rank_ms_per_step"])
    assert out["standin"] == "tests.standin_eu" and "NOT a measurement" in out["data"]
    if cfg[1] == "c5":
        assert out["gather"]["ms"] > 0 and out["verified"]["max_rel_err"] <= 1e-12
    if cfg[1] == "c3":
        assert out["verified"]["replicas_bitwise_equal"] is True and out["config"]["n"] == 101
    if cfg[1] == "c2":
        assert out["scaling"] == "weak" and out["value"] > 0


def test_bench_refuses_the_standin_outside_the_tests():
    """--standin swaps the product for an oracle-backed CPU solver inside the bench harness; a flag alone must not be
    enough to put its output into a bench line."""
    r, out, lines = _run_bench(["--gpus", "1", "--config", "c2", "--n", "300"], env_extra={"EXPV_MI_BENCH_STANDIN_OK": ""})
    assert r.returncode != 0 and not lines and "refusing" in (r.stderr + r.stdout)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """One rank per GPU is the contract: `--gpus 2` inside a 1-rank job must fail loudly, not print an n_gpus: 1 line."""
    r, out, lines = _run_bench(["--gpus", "2", "--config", "c2", "--n", "300"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_rank_check_rejects_ranks_sharing_a_device():
    import argparse
    import sys
    sys.path.insert(0, ROOT)
    import bench

    class _Env(bench.Env):
        def device_ids(self):
            return ["host/gpu0", "host/gpu0"]

        def ranks_seen(self):
            return 2
    with pytest.raises(SystemExit) as ei:
        _Env(torch, dist, 2, 0, "cpu", None).check_ranks(2)
    assert "distinct" in str(ei.value)


def test_bench_result_line_stays_under_4k_for_the_driver():
    """BENCH_r04.json had `parsed: null` because the one JSON line had grown to 21.7 KB.  The LAST stdout line is
    now a compact record (headline keys, flat config / roofline / cpu_baseline); the full record goes to bench_full.json.  Checked
    here on the very record that broke the driver (profiles/r04_bench_final.json) and on an 8-rank worst case."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_final.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) <= 1e-8 * full["value"] and abs(d["value"] - 30 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert "kernels" not in d["roofline"] and "note" not in d["roofline"] and "secondary" not in d
    assert set(d["config"]["secondary_fracs"]) >= {"lanczos", "c4_kiops_complex", "general_sparse_random"}
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert len(d["cpu_baseline"]["sample"]) <= 120 and len(d["roofline"]["traffic_source"]) <= 80
    # 8 ranks, twice as many secondaries with long names: the line sheds optional parts instead of growing past the limit
    big = dict(full, n_gpus=8, ranks_seen=8, devices=["runc/%032d/pci-0:%d.0" % (i, i) for i in range(8)],
               per_rank_ms_per_step=[1.1153013 + 1e-3 * i for i in range(8)])
    big["config"] = dict(full["config"])
    big["config"]["secondary_fracs"] = {("a_rather_long_secondary_entry_name_%03d" % i): 0.123456 for i in range(120)}
    line8 = bench.compact_line(big)
    assert len(line8) < 4096
    d8 = json.loads(line8)
    assert d8["n_gpus"] == 8 and d8["roofline"]["frac"] == d["roofline"]["frac"] and "cpu_baseline" in d8

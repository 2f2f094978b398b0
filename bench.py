"""bench.py -- headline benchmark: expv matvecs/s on BASELINE config 2
(n = 1e6, 5-diagonal non-symmetric sparse fp64, m = 30, t = 1.0; SURVEY.md §8d inputs).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full expv(t, A, b): firststep + 30 Krylov steps (operator apply +
orthogonalisation + normalisation) + host Pade of the 30x30 Hessenberg + the beta*V*coef combine,
with A, b and w resident in HBM.  One unit of the metric = one Krylov step ("matvec"), so an expv
at m = 30 is 30 units.  Every rank runs its own independent (A, b) problem (weak scaling, no
data-path collective); value = N * K * 30 / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ROWS = 1_000_000
M_KRYLOV = 30
T_FINAL = 1.0
HBM_PEAK_GBS = 8000.0


def c2_operator(n):
    from tests._util import c2_operator as mk
    return mk(n)


def alg_bytes_expv(n, nnz, m, s=8):
    """SURVEY.md §8d contract figure: m*A_B + s*n*(m(m+1)/2 + 3m + 3), A_B = nnz*(s+4) + 4(n+1)."""
    a_b = nnz * (s + 4) + 4 * (n + 1)
    return m * a_b + s * n * (m * (m + 1) // 2 + 3 * m + 3)


def alg_bytes_kernel(name, n, nnz, m, s=8):
    """Average algorithmic bytes of ONE launch of a kernel over the m steps of an expv."""
    a_b = nnz * (s + 4) + 4 * (n + 1)
    avg_j = (m + 1) / 2.0
    return {
        "matvec": a_b + 2 * s * n,                 # read A, read x, write y
        "dots": s * n * (avg_j + 2),               # read V[:,1:j], y and v_j (Gram row)
        "update": s * n * (avg_j + 2),             # read V[:,1:j] and y, write y
        "scale": 2 * s * n,                        # read y, write v_{j+1}
        "combine": s * n * (m + 1),                # read V[:,1:m], write w
        "firststep": 3 * s * n / 2.0,              # sumsq reads b; scale_copy reads b, writes v_1 (2 launches)
        "fused_a": a_b + s * n * (avg_j + 2),      # A + x + V[:,1:j-1] read, v_j and y written
        "fused_b": s * n * (avg_j + 2),
        # banded pipeline: ONE launch = one whole Krylov step = the contract's per-matvec figure
        # A_B + s*n*(j + 3) (A, x, y, the window read for the projections and again for the
        # update), averaged over j = 1..m.  The kernel itself moves less (window read once, DIA diagonals
        # without column indices): "traffic" below is what it really moved.
        "pipe_step": a_b + s * n * (avg_j + 3),
    }.get(name)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate runs, gfx950 x2 read correction applied: tools/pmc_summary.py).  PMC collection
    cannot run inside this process, so the per-launch figure measured on the same command is read from
    profiles/; None when no pass has been recorded for this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files or kernel is None:
        return None
    data = json.load(open(files[-1]))
    key = {"fused_a": "k_fused_a", "fused_b": "k_update2", "dots": "k_dots", "update": "k_update<",
           "matvec": "k_spmv", "combine": "k_combine", "pipe_step": "k_pipe<"}.get(kernel)
    if key is None:
        return None
    tot_b = tot_n = 0.0
    for name, v in data.items():
        if name.startswith(key):
            tot_b += v["hbm_bytes_per_launch"] * v["launches"]
            tot_n += v["launches"]
    return (tot_b / tot_n) if tot_n else None


def run_c5(args, eu, ctx, world, rank, dist, torch):
    """BASELINE configs[4]: nprob independent expv problems (n = 1e5, C2 diagonals scaled per problem,
    m = 30), problems sharded over the ranks, one final gather of the results."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi_dist", os.path.join(ROOT, "exponentialutilities.jl_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    n, m, nprob = 100_000, M_KRYLOV, args.nprob
    A0 = c2_operator(n).tocsr()
    A0.sort_indices()
    nnz = A0.nnz
    lo, hi = D.shard_range(nprob, world, rank)
    rng = np.random.default_rng(7)
    scales = 1 + 0.1 * rng.random(nprob)
    vals = torch.as_tensor(np.stack([A0.data * s for s in scales[lo:hi]]), device="cuda")
    B = torch.as_tensor(np.random.default_rng(100 + rank).standard_normal((hi - lo, n)), device="cuda").t()
    def step():
        W = eu.expv_batch(T_FINAL, A0, vals, B, m=m, ctx=ctx)
        return D.gather_columns(W, nprob) if world > 1 else W
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); ctx.sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        W = step()
    torch.cuda.synchronize(); ctx.sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    units, elapsed = D.aggregate_throughput((hi - lo) * m * args.steps, elapsed, device="cuda")
    b_alg = alg_bytes_expv(n, nnz, m) * nprob
    out = {"metric": "expv matvecs/s, batch of independent problems n=1e5 sparse fp64 m=30", "value": units / elapsed,
           "unit": "matvecs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4]: %d independent expv, n=1e5, 5-diagonal, m=30, sharded over %d "
                                  "GPU(s), final gather" % (nprob, world), "nprob": nprob, "n": n, "m": m},
           "roofline": {"bound": "hbm", "achieved": b_alg / (elapsed / args.steps) / 1e9 / world, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": b_alg / (elapsed / args.steps) / 1e9 / world / HBM_PEAK_GBS,
                        "traffic": None, "note": "whole-call algorithmic GB/s per GPU (V of a problem is cache-resident)"}}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_ROWS, help="override problem size (debug only; invalidates the metric)")
    ap.add_argument("--ortho", default="auto", choices=["auto", "mgs", "lowsync"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra non-overlapped profiling pass")
    ap.add_argument("--sync-outputs", action="store_true", help="every call returns only when its device result is complete")
    ap.add_argument("--split-api", action="store_true", help="time arnoldi!(Ks,A,b) + expv!(w,t,Ks) instead of expv(t,A,b)")
    ap.add_argument("--config", default="c2", choices=["c2", "c5"],
                    help="c2 (default, the headline metric) or c5: batch of --nprob independent n=1e5 problems")
    ap.add_argument("--nprob", type=int, default=1024, help="c5: total number of problems over all GPUs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, (world, args.gpus)

    import expv_mi_loader
    eu = expv_mi_loader.load()
    # device-resident results are stream-ordered (like any HIP library); barrier() below syncs the context
    ctx = eu.Context(device=local_rank, async_outputs=not args.sync_outputs)
    if args.config == "c5":
        return run_c5(args, eu, ctx, world, rank, dist, torch)
    n, m = args.n, M_KRYLOV
    A = c2_operator(n)
    nnz = A.nnz
    t_setup = time.perf_counter()
    op = eu.MIOperator(A, ctx)                      # CSR32 upload + properties: setup, not timed
    t_setup = time.perf_counter() - t_setup
    b_host = np.random.default_rng(3 + rank).standard_normal(n)
    b = torch.as_tensor(b_host, device="cuda")
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)

    def one_expv():
        if args.split_api:      # arnoldi!(Ks, A, b) then expv!(w, t, Ks): the two-call form of the reference
            eu.arnoldi_(Ks, op, b, m=m, ishermitian=False, ortho=args.ortho)
            eu.expv_(w, T_FINAL, Ks)
            return Ks.m
        eu.expv(T_FINAL, op, b, m=m, ishermitian=False, ortho=args.ortho, out=w)    # expv(t, A, b; m)
        return eu.expv.last_stats["m"]

    def barrier():
        torch.cuda.synchronize()
        ctx.sync()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        one_expv()
    barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += one_expv()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        uu = torch.tensor([units], dtype=torch.float64, device="cuda")
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
        units_total = float(uu.item())
    else:
        units_total = float(units)
    value = units_total / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- per-kernel HIP-event timing of the same K steps (separate pass: events perturb the headline) ---
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(args.steps):
        one_expv()
    ctx.sync()
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    # the banded pipeline records its step kernel under "fused_a" and has no per-step "fused_b"
    pipeline = "fused_a" in prof and prof.get("fused_b", {"launches": 0})["launches"] < prof["fused_a"]["launches"] / 2
    if pipeline:
        prof["pipe_step"] = prof.pop("fused_a")
    serial = None
    if pipeline and not args.no_serial_pass:
        # the default mode overlaps consecutive step kernels (two streams), so a kernel has no duration of its own:
        # "avg_ms" above is the factorisation span / steps.  One more pass with one launch after the other gives
        # per-launch HIP-event durations that a rocprofv3 --kernel-trace of EXPV_MI_PIPE_SERIAL=1 reproduces.
        ctx.set_pipeline_overlap(False)
        for _ in range(2):
            one_expv()
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_expv()
        ctx.sync()
        t_serial = (time.perf_counter() - t1) / args.steps
        ps = ctx.prof_get()
        ctx.prof_enable(False)
        ctx.set_pipeline_overlap(True)
        if "fused_a" in ps:
            avg = ps["fused_a"]["total_ms"] / ps["fused_a"]["launches"]
            ab = alg_bytes_kernel("pipe_step", n, nnz, m)
            serial = {"avg_launch_ms": avg, "alg_GBps": ab / (avg * 1e-3) / 1e9, "frac": ab / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "ms_per_expv": 1e3 * t_serial, "matvecs_per_s": m / t_serial}
    kern = {}
    for name, p in prof.items():
        ab = alg_bytes_kernel(name, n, nnz, m)
        avg_ms = p["total_ms"] / p["launches"]
        kern[name] = {"launches_per_expv": p["launches"] / args.steps, "avg_ms": avg_ms,
                      "total_ms_per_expv": p["total_ms"] / args.steps,
                      "alg_GBps": (ab / (avg_ms * 1e-3) / 1e9) if ab else None}
    dom = max(kern, key=lambda k: kern[k]["total_ms_per_expv"]) if kern else None
    traffic = pmc_traffic(dom) if n == N_ROWS else None
    b_alg = alg_bytes_expv(n, nnz, m)
    expv_gbps = b_alg / (elapsed / args.steps) / 1e9
    roofline = {
        "bound": "hbm", "kernel": dom,
        "achieved": kern[dom]["alg_GBps"] if dom else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (kern[dom]["alg_GBps"] / HBM_PEAK_GBS) if dom and kern[dom]["alg_GBps"] else None,
        "traffic": traffic,
        "avg_launch_ms": kern[dom]["avg_ms"] if dom else None,
        "expv_alg_GBps": expv_gbps, "expv_frac": expv_gbps / HBM_PEAK_GBS,
        "kernels": kern,
    }
    if pipeline:
        roofline["note"] = ("pipe_step = k_pipe_live/k_pipe, one launch per Krylov step; consecutive launches overlap "
                            "(two streams), avg_launch_ms = factorisation span / steps; 'serial' = same kernels one "
                            "after the other (per-launch HIP events)")
        roofline["serial"] = serial

    out = {
        "metric": "expv matvecs/s (Krylov steps/s), n=1e6 5-diagonal sparse fp64, m=30",
        "value": value, "unit": "matvecs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: expv(1.0, A, b), n=%d, offsets (-2..2) diagonals "
                               "(0.3,1.2,-2.0,0.8,-0.1), nnz=%d, m=30, tol=1e-7, full Arnoldi (%s), "
                               "one independent problem per GPU" % (n, nnz, args.ortho),
                   "n": n, "m": m, "nnz": int(nnz), "ortho": args.ortho, "setup_s": t_setup},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the plain-C restatement of the reference loop (literal MGS), same workload
        from oracle import c_oracle as co
        threads = co.num_threads()
        reps, tc = 0, 0.0
        t_start = time.perf_counter()
        while reps < 3 and (time.perf_counter() - t_start) < 25.0:
            t1 = time.perf_counter()
            wo, r = co.expv_csr(T_FINAL, A, b_host, m=m)
            tc += time.perf_counter() - t1
            reps += 1
        cpu_val = reps * r["m"] / tc
        err = float(np.linalg.norm(w.cpu().numpy() - wo) / np.linalg.norm(wo))
        out["cpu_baseline"] = {"value": cpu_val, "unit": "matvecs/s", "cores": threads, "kind": "port",
                               "sample": "%d full expv calls of the same workload (n=%d, m=30) through "
                                         "oracle/expv_oracle.c, OpenMP threads=%d" % (reps, n, threads),
                               "parity_rel_err_w": err}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Secondary measurements (not the headline metric): BASELINE configs 3 and 4 on one MI355X.
  c3: phiv_timestep adaptive (K=4 phi functions), dense fp64 A (n chosen to fit quickly; 8n^2 B per matvec)
  c4: kiops, n=1e6 sparse complex-fp64, iop=2 (complex = this build's extension, no reference behaviour)
Prints one JSON line per config."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import expv_mi_loader
from tests._util import c2_operator

eu = expv_mi_loader.load()
ctx = eu.default_context()
which = sys.argv[1:] or ["c3", "c4", "c2sym", "c2stencil"]

if "c3" in which:
    for n in (16384, 65536):
        g = torch.Generator(device="cuda").manual_seed(4)
        A = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g) / np.sqrt(n)
        A.diagonal().add_(-2.0)
        op = eu.MIOperator(A.t())            # column-major view of the same memory (the transpose of a row-major tensor)
        B = torch.randn(5, n, dtype=torch.float64, device="cuda", generator=g).t()
        st = {}
        eu.phiv_timestep(1.0, op, B, adaptive=True, tol=1e-7, m=10, stats=st)      # warm-up
        ctx.sync()
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        u = eu.phiv_timestep(1.0, op, B, adaptive=True, tol=1e-7, m=10, stats=st)
        ctx.sync()
        dt = time.perf_counter() - t0
        prof = ctx.prof_get(); ctx.prof_enable(False)
        mv = prof.get("matvec", {"launches": 1, "total_ms": float("nan")})
        gemv_gbps = 8.0 * n * n / (mv["total_ms"] / mv["launches"] * 1e-3) / 1e9
        print(json.dumps({"config": "c3 phiv_timestep adaptive K=4 dense fp64", "n": n, "seconds": dt, "matvecs": st["matvecs"],
                          "matvecs_per_s": st["matvecs"] / dt, "num_timesteps": st["num_timesteps"], "m_final": st["m"],
                          "gemv_avg_ms": mv["total_ms"] / mv["launches"], "gemv_alg_GBps": gemv_gbps,
                          "gemv_frac_of_8TBps": gemv_gbps / 8000.0}))
        del A, op, B, u

if "c4" in which:
    n = 1_000_000
    A = (c2_operator(n) * (1 + 0.25j)).tocsc()
    u = np.random.default_rng(6).standard_normal(n) + 1j * np.random.default_rng(60).standard_normal(n)
    op = eu.MIOperator(A)
    ud = torch.as_tensor(u, device="cuda")
    w, st = eu.kiops(1.0, op, ud, allow_complex=True, ishermitian=False, mmin=10, mmax=128)
    ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    w, st = eu.kiops(1.0, op, ud, allow_complex=True, ishermitian=False, mmin=10, mmax=128)
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.prof_get(); ctx.prof_enable(False)
    steps = prof.get("matvec", {"launches": 0})["launches"] + prof.get("fused_a", {"launches": 0})["launches"]   # modular + fused steps
    print(json.dumps({"config": "c4 kiops n=1e6 sparse complex-fp64 iop=2", "seconds": dt, "stats": st,
                      "krylov_steps": steps, "steps_per_s": steps / dt,
                      "alg_MB_per_step": 184, "frac_of_8TBps": 184e6 * steps / dt / 8e12,
                      "kernels": {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in prof.items()}}))


def _c2_variant(name, A, herm, contract_bytes_per_step):
    """SURVEY.md §8d secondary inputs on the C2 size: whole-call expv, m = 30."""
    import scipy.sparse as sp
    n, m = A.shape[0], 30
    actx = eu.Context(async_outputs=True)
    op = eu.MIOperator(A, actx)
    b = torch.as_tensor(np.random.default_rng(3).standard_normal(n), device="cuda")
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        eu.expv(1.0, op, b, m=m, ishermitian=herm, out=w)
    actx.sync()
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        eu.expv(1.0, op, b, m=m, ishermitian=herm, out=w)
    actx.sync()
    dt = (time.perf_counter() - t0) / reps
    mm = eu.expv.last_stats["m"]
    print(json.dumps({"config": name, "n": n, "m": mm, "ms_per_expv": 1e3 * dt, "matvecs_per_s": mm / dt,
                      "contract_MB_per_step": contract_bytes_per_step / 1e6,
                      "frac_of_8TBps": contract_bytes_per_step * mm / dt / 8e12}))


if "c2sym" in which:      # symmetric diagonals (0.5, 1, -3, 1, 0.5): Lanczos path, window of 2
    import scipy.sparse as sp
    n = 1_000_000
    A = sp.diags([0.5, 1.0, -3.0, 1.0, 0.5], [-2, -1, 0, 1, 2], shape=(n, n), format="csc")
    _c2_variant("c2 symmetric (Lanczos), n=1e6 5-diagonal", A, True, A.nnz * 12 + 4 * (n + 1) + 8 * n * 4)

if "c2stencil" in which:  # 2-D 5-point stencil offsets (-1000, -1, 0, 1, 1000): non-local x access, two-kernel path
    import scipy.sparse as sp
    n = 1_000_000
    A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-1000, -1, 0, 1, 1000], shape=(n, n), format="csc")
    _c2_variant("c2 stencil offsets (-1000,-1,0,1,1000), full Arnoldi", A, False, A.nnz * 12 + 4 * (n + 1) + 8 * n * (15.5 + 3))

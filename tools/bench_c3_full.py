"""BASELINE config 3 at the largest size that fits one MI355X: phiv_timestep adaptive (K = 4), dense fp64 A,
n = 163 840 (214.7 GB; the literal n = 2e5 is 320 GB > 288 GB HBM).  Prints one JSON line."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
eu = expv_mi_loader.load()
ctx = eu.default_context()
n = 163_840
g = torch.Generator(device="cuda").manual_seed(4)
A = torch.empty(n, n, dtype=torch.float64, device="cuda")
for r0 in range(0, n, 8192):      # fill in slabs: randn of the whole matrix would need a second 215 GB buffer
    A[r0:r0 + 8192].copy_(torch.randn(8192, n, dtype=torch.float64, device="cuda", generator=g) / np.sqrt(n))
A.diagonal().add_(-2.0)
op = eu.MIOperator(A.t())
B = torch.randn(5, n, dtype=torch.float64, device="cuda", generator=g).t()
st = {}
eu.phiv_timestep(1.0, op, B, adaptive=True, tol=1e-7, m=10, stats=st)
ctx.sync()
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter()
u = eu.phiv_timestep(1.0, op, B, adaptive=True, tol=1e-7, m=10, stats=st)
ctx.sync()
dt = time.perf_counter() - t0
prof = ctx.prof_get()
mv = prof["matvec"]
gb = 8.0 * n * n / (mv["total_ms"] / mv["launches"] * 1e-3) / 1e9
print(json.dumps({"config": "c3 phiv_timestep adaptive K=4 dense fp64, n=163840 (214.7 GB)", "seconds": dt, "matvecs": st["matvecs"],
                  "matvecs_per_s": st["matvecs"] / dt, "gemv_avg_ms": mv["total_ms"] / mv["launches"], "gemv_alg_GBps": gb,
                  "gemv_frac_of_8TBps": gb / 8000.0}))

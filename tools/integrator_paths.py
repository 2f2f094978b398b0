import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
B = torch.randn(n, 3, dtype=torch.float64, device="cuda")
def timeit(f, reps=10):
    for _ in range(3): f()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return 1e3 * (time.perf_counter() - t0) / reps
c0 = ctx.counters()
print("expv m=30           ms", timeit(lambda: eu.expv(1.0, op, b, m=30, ishermitian=False)))
c1 = ctx.counters(); print("  steps/call", (c1["krylov_steps"] - c0["krylov_steps"]) / 13)
for tol in (1e-6, 1e-10):
    c0 = ctx.counters()
    t = timeit(lambda: eu.expv_timestep([0.5, 1.0], op, b, tol=tol, adaptive=True))
    c1 = ctx.counters(); st = (c1["krylov_steps"] - c0["krylov_steps"]) / 13
    print("expv_timestep adaptive tol", tol, "ms", t, "steps/call", st, "us/step", 1e3 * t / st, "op_applies", (c1["op_applies"] - c0["op_applies"]) / 13)
c0 = ctx.counters()
t = timeit(lambda: eu.phiv_timestep([1.0], op, B, tol=1e-8, adaptive=True))
c1 = ctx.counters(); st = (c1["krylov_steps"] - c0["krylov_steps"]) / 13
print("phiv_timestep K=2 adaptive ms", t, "steps/call", st, "us/step", 1e3 * t / st, "op_applies", (c1["op_applies"] - c0["op_applies"]) / 13)
c0 = ctx.counters()
t = timeit(lambda: eu.kiops(1.0, op, B, tol=1e-8))
c1 = ctx.counters(); st = (c1["krylov_steps"] - c0["krylov_steps"]) / 13
print("kiops real K=2 ms", t, "steps/call", st, "us/step", 1e3 * t / st)

from tests._util import c2_operator as c2u
As = eu.MIOperator(c2u(n, sym=True), ctx)
def count(f, label):
    c0 = ctx.counters(); t = timeit(f); c1 = ctx.counters(); st = (c1["krylov_steps"] - c0["krylov_steps"]) / 13
    print(label, "ms", round(t, 4), "steps/call", st, "us/step", round(1e3 * t / max(st, 1), 1))
count(lambda: eu.phiv(1.0, op, b, 3, m=30), "phiv k=3 m=30")
if As is not None:
    count(lambda: eu.expv(1.0, As, b, m=30), "expv hermitian (Lanczos)")
    count(lambda: eu.expv(1.0, As, b, m=30, mode="error_estimate", rtol=1e-8), "expv hermitian error_estimate")
    count(lambda: eu.expv_timestep([1.0], As, b, tol=1e-8, adaptive=True), "expv_timestep hermitian adaptive")
Ks = eu.arnoldi(op, b, m=30)
count(lambda: eu.arnoldi_(Ks, op, b, m=30) if hasattr(eu, "arnoldi_") else eu.arnoldi(op, b, m=30), "arnoldi! into an existing Ks")
count(lambda: eu.arnoldi(op, b, m=30), "arnoldi (fresh Ks per call)")

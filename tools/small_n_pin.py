"""small_n.py with the calling thread pinned to one CPU after the library is up (PIN=cpu index within the allowed set)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mode = os.environ.get("PIN", "")
if mode == "early":
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[2]})
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
for _ in range(5):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
if mode == "late":
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[2]})
t0 = time.perf_counter()
for _ in range(reps):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
print({"n": n, "ms_per_expv": round(1e3 * (time.perf_counter() - t0) / reps, 4), "pin": mode, "omp": os.environ.get("OMP_NUM_THREADS")})

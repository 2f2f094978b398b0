"""BASELINE configs[3] (kiops, n = 1e6 complex sparse, iop = 2) a few times: the command rocprofv3 traces for
profiles/rNN_c4_*.  Prints wall time per call and the context counters."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import expv_mi_loader
from bench import c2_operator

eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1_000_000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A = (c2_operator(n) * (1 + 0.25j)).tocsc()
op = eu.MIOperator(A, ctx)
rng = np.random.default_rng(6)
u = torch.as_tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n), device="cuda")
for _ in range(2):
    w, st = eu.kiops(1.0, op, u, allow_complex=True, ishermitian=False, opnorm=4.6)
ctx.sync()
c0 = ctx.counters()
t0 = time.perf_counter()
for _ in range(reps):
    w, st = eu.kiops(1.0, op, u, allow_complex=True, ishermitian=False, opnorm=4.6)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
c1 = ctx.counters()
print({"ms_per_call": 1e3 * dt, "stats": st, "krylov_steps_per_call": (c1["krylov_steps"] - c0["krylov_steps"]) / reps})

"""BASELINE configs[4] share of one GPU (128 problems, n = 1e5, m = 30) a few times: the command rocprofv3 traces."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n, m, nprob = 100_000, 30, int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A0 = c2_operator(n).tocsr(); A0.sort_indices()
scales = 1 + 0.1 * np.random.default_rng(7).random(nprob)
vals = torch.as_tensor(np.stack([A0.data * s for s in scales]), device="cuda")
B = torch.as_tensor(np.random.default_rng(1).standard_normal((nprob, n)), device="cuda").t()
for _ in range(2):
    W = eu.expv_batch(1.0, A0, vals, B, m=m, ctx=ctx)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    W = eu.expv_batch(1.0, A0, vals, B, m=m, ctx=ctx)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print({"ms_per_call": 1e3 * dt, "matvecs_per_s": nprob * m / dt})

"""expv_batch at other shapes than config 5's (128 x 1e5): share of the HBM roofline by the same contract bytes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
m = 30
for n, nprob in ((10_000, 1024), (30_000, 512), (100_000, 128), (300_000, 48), (1_000_000, 12)):
    A0 = c2_operator(n).tocsr(); A0.sort_indices()
    scales = 1 + 0.1 * np.random.default_rng(7).random(nprob)
    vals = torch.as_tensor(np.stack([A0.data * s for s in scales]), device="cuda")
    B = torch.as_tensor(np.random.default_rng(1).standard_normal((nprob, n)), device="cuda").t()
    for _ in range(2):
        W = eu.expv_batch(1.0, A0, vals, B, m=m, ctx=ctx)
    ctx.sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        W = eu.expv_batch(1.0, A0, vals, B, m=m, ctx=ctx)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    nnz = A0.nnz
    bytes_per = m * (12 * nnz + 4 * (n + 1)) + 8 * n * (m * (m + 1) // 2 + 3 * m + 3)
    print("n %8d x %5d problems: %8.3f ms per call, %8.1f k matvecs/s, %.3f of 8 TB/s" % (n, nprob, 1e3 * dt, nprob * m / dt / 1e3, nprob * bytes_per / dt / 8e12))
    del vals, B, W

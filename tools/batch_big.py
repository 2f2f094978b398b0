"""expv_batch on a few LARGE problems (n = 1e6): the command rocprofv3 traces for the batch timeline.  usage: python tools/batch_big.py [nprob]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n, nprob, m = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 12, 30
A0 = c2_operator(n).tocsr(); A0.sort_indices()
vals = torch.as_tensor(np.stack([A0.data * (1 + 0.01 * s) for s in range(nprob)]), device="cuda")
B = torch.as_tensor(np.random.default_rng(1).standard_normal((nprob, n)), device="cuda").t()
for _ in range(3):
    W = eu.expv_batch(1.0, A0, vals, B, m=m, ctx=ctx)
ctx.sync()

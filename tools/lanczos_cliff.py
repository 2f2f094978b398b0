import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
for n in (50000, 90000, 100000, 110000, 200000, 500000, 1000000):
    A = c2_operator(n, sym=True)
    ctx = eu.Context(async_outputs=True)
    op = eu.MIOperator(A, ctx)
    bt = torch.randn(n, dtype=torch.float64, device="cuda")
    wt = torch.empty_like(bt)
    for _ in range(5):
        eu.expv(1.0, op, bt, m=30, ishermitian=True, out=wt)
    ctx.sync()
    c0 = ctx.counters()
    t0 = time.perf_counter()
    for _ in range(20):
        eu.expv(1.0, op, bt, m=30, ishermitian=True, out=wt)
    ctx.sync()
    dt = 1e3 * (time.perf_counter() - t0) / 20
    c1 = ctx.counters()
    print(n, "%.3f ms" % dt, eu.expv.last_stats["path"], {k: c1[k] - c0[k] for k in c1})

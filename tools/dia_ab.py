import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader, bench
eu = expv_mi_loader.load()
n=1_000_000
for dia in (1,0):
    ctx = eu.Context(async_outputs=True)
    ctx.set_option("dia", dia)
    op = eu.MIOperator(bench.c2_operator(n), ctx)
    b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
    f = lambda: eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
    for _ in range(5): f()
    ctx.sync(); ts=[]
    for _ in range(5):
        t0=time.perf_counter()
        for _ in range(20): f()
        ctx.sync(); ts.append((time.perf_counter()-t0)/20)
    t=sorted(ts)[2]
    print("dia=%d: %.3f ms, %.3f of contract, path %s" % (dia, 1e3*t, 6.384e9/t/8e12, eu.expv.last_stats["path"]))

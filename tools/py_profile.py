import sys, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader, bench
eu = expv_mi_loader.load()
n = 1_000_000
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(bench.c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
f = lambda: eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
for _ in range(20): f()
ctx.sync()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f()
ctx.sync(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)

import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n, m = 1_000_000, 30
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
for _ in range(5): f()
torch.cuda.synchronize()
# python-side cost: profile 200 calls
pr = cProfile.Profile(); pr.enable()
for _ in range(200): f()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(18)

import sys
sys.path.insert(0, ".")
import numpy as np, torch, time
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1000000
op = eu.MIOperator(c2_operator(n, sym=True), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
Ks = eu.KrylovSubspace(np.float64, np.float64, n, 30, 0, ctx)
def t_of(f, reps=20):
    f(); ctx.sync()
    return 1e3 * timed(f, reps, 2, ctx.sync)
print("fresh m=10          ms %.3f" % t_of(lambda: eu.arnoldi_(Ks, op, b, m=10, ishermitian=True, tol=-1.0)))
print("fresh m=16          ms %.3f" % t_of(lambda: eu.arnoldi_(Ks, op, b, m=16, ishermitian=True, tol=-1.0)))
print("fresh m=30          ms %.3f" % t_of(lambda: eu.arnoldi_(Ks, op, b, m=30, ishermitian=True, tol=-1.0)))
def two():
    eu.arnoldi_(Ks, op, b, m=10, ishermitian=True, tol=-1.0)
    eu.arnoldi_(Ks, op, b, m=16, ishermitian=True, tol=-1.0, init=10)
print("m=10 then init=10 -> 16   ms %.3f" % t_of(two))
c0 = ctx.counters(); two(); ctx.sync(); c1 = ctx.counters()
print({k: c1[k] - c0[k] for k in c0})
def three():
    eu.arnoldi_(Ks, op, b, m=10, ishermitian=True, tol=-1.0)
    eu.arnoldi_(Ks, op, b, m=16, ishermitian=True, tol=-1.0, init=10)
    eu.arnoldi_(Ks, op, b, m=22, ishermitian=True, tol=-1.0, init=16)
print("10 -> 16 -> 22      ms %.3f" % t_of(three))

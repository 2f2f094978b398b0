"""Wall-time split of one expv on config 2: arnoldi! alone vs arnoldi! + expv! (host Pade + combine)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load(); ctx = eu.default_context()
n, m = 1_000_000, 30
A = c2_operator(n); op = eu.MIOperator(A)
b = torch.as_tensor(np.random.default_rng(3).standard_normal(n), device="cuda")
w = torch.empty(n, dtype=torch.float64, device="cuda")
Ks = eu.KrylovSubspace(np.float64, np.float64, n, m)
def t(fn, reps=20):
    for _ in range(3): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps * 1e6
a = t(lambda: eu.arnoldi_(Ks, op, b, m=m, ishermitian=False))
e = t(lambda: eu.expv_(w, 1.0, Ks))
both = t(lambda: (eu.arnoldi_(Ks, op, b, m=m, ishermitian=False), eu.expv_(w, 1.0, Ks)))
print({"arnoldi_us": round(a, 1), "expv_only_us": round(e, 1), "both_us": round(both, 1)})

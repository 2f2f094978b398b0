"""What the convenience forms cost per call at n = 1e6 (config-2 operator): the matrix passed as a scipy / numpy object on every
call (fingerprint check, cached upload), operator creation, host vectors."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
A = c2_operator(n)
ctx = eu.default_context() if hasattr(eu, "default_context") else eu.Context()
b = torch.randn(n, dtype=torch.float64, device="cuda")
bh = b.cpu().numpy()
def timeit(f, reps=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0) / reps, 3)
t0 = time.perf_counter(); op = eu.MIOperator(A); torch.cuda.synchronize(); print("MIOperator(A) creation ms", round(1e3 * (time.perf_counter() - t0), 2))
t0 = time.perf_counter(); op2 = eu.MIOperator(A); torch.cuda.synchronize(); print("MIOperator(A) creation again ms", round(1e3 * (time.perf_counter() - t0), 2))
print("expv(t, op, b_dev)            ms", timeit(lambda: eu.expv(1.0, op, b, m=30, ishermitian=False)))
print("expv(t, A_scipy, b_dev)       ms", timeit(lambda: eu.expv(1.0, A, b, m=30, ishermitian=False)))
print("expv(t, op, b_host)           ms", timeit(lambda: eu.expv(1.0, op, bh, m=30, ishermitian=False)))
print("expv(t, A_scipy, b_host)      ms", timeit(lambda: eu.expv(1.0, A, bh, m=30, ishermitian=False)))
print("expv(t, op, b_dev) ishermitian=None ms", timeit(lambda: eu.expv(1.0, op, b, m=30)))
Ad = A.tocsc()
print("expv(t, A_csc, b_dev)         ms", timeit(lambda: eu.expv(1.0, Ad, b, m=30, ishermitian=False)))

"""Typical user-level calls with HOST (numpy) operands at n = 1e6: looking for cliffs next to the device-resident figures."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
A = c2_operator(n).tocsc(); A.sort_indices()
op = eu.MIOperator(A)
rng = np.random.default_rng(0)
b = rng.standard_normal(n)
B = rng.standard_normal((n, 3))                 # row-major, as numpy makes it
Bf = np.asfortranarray(B)
def timeit(f, label, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); print("%-58s %8.3f ms" % (label, 1e3 * (time.perf_counter() - t0) / reps))
timeit(lambda: eu.expv(1.0, op, b, m=30, ishermitian=False), "expv, numpy b")
timeit(lambda: eu.expv(1.0j, op, b, m=30, ishermitian=False), "expv, imaginary t (complex result), numpy b")
timeit(lambda: eu.phiv(1.0, op, b, 3, m=30), "phiv k = 3, numpy b")
timeit(lambda: eu.kiops(1.0, op, B, tol=1e-8), "kiops, numpy u (n x 3 row-major)")
timeit(lambda: eu.kiops(1.0, op, Bf, tol=1e-8), "kiops, numpy u (n x 3 column-major)")
timeit(lambda: eu.phiv_timestep([1.0], op, B, tol=1e-8, adaptive=True), "phiv_timestep adaptive, numpy B (row-major)")
timeit(lambda: eu.phiv_timestep([1.0], op, Bf, tol=1e-8, adaptive=True), "phiv_timestep adaptive, numpy B (column-major)")
Ac = (A * (1.0 + 0.2j)).tocsc(); Ac.sort_indices()
t0 = time.perf_counter(); opc = eu.MIOperator(Ac); torch.cuda.synchronize(); print("%-58s %8.3f ms" % ("complex MIOperator creation", 1e3 * (time.perf_counter() - t0)))
bc = b + 1j * rng.standard_normal(n)
timeit(lambda: eu.expv(1.0, opc, bc, m=14, ishermitian=False), "complex expv m = 14, numpy b")

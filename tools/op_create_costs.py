"""Creation time of a sparse operator (n = 1e6, nnz = 5e6) from CSR / CSC / DIA-format scipy matrices and of the Python-side conversions;
with EXPV_MI_OP_TIMING=1 the library prints the phases of each build."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
A = c2_operator(n)
print("format", A.format)
def T(f, label, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, "ms", round(1e3 * min(ts), 2)); return r
Ac = T(lambda: A.tocsc(), "tocsc")
Ac2 = T(lambda: Ac.astype(np.float64), "astype")
T(lambda: Ac2.sort_indices(), "sort_indices (already sorted)")
T(lambda: (np.ascontiguousarray(Ac2.indptr, dtype=np.int64), np.ascontiguousarray(Ac2.indices, dtype=np.int64)), "index widen")
Ar = Ac.tocsr(); Ar.sort_indices()
op = T(lambda: eu.MIOperator(Ar), "MIOperator(csr)")
op = T(lambda: eu.MIOperator(Ac), "MIOperator(csc)")
op = T(lambda: eu.MIOperator(A), "MIOperator(dia-format scipy)")
import os
os.environ["EXPV_MI_HOST_TIMING"] = "1"

import sys
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import expv_mi_loader
from oracle import krylov_oracle as ko
eu = expv_mi_loader.load()
n, m = 20000, 20
rng = np.random.default_rng(1)
rows = np.repeat(np.arange(n), 5)
cols = rng.integers(0, n, 5 * n)
A = sp.csr_matrix((rng.standard_normal(5 * n) * 0.3, (rows, cols)), shape=(n, n)) - 0.5 * sp.eye(n)
A = A.tocsr(); A.sum_duplicates()
b = rng.standard_normal(n)
want = ko.expv(1.0, A, b, m=m, ishermitian=False)
for opt in (0, 1):
    ctx = eu.Context()
    ctx.set_option("fa2_pipelined", opt)
    ctx.set_option("pipeline", 0)
    ctx.set_option("reorder", 0)
    op = eu.MIOperator(A, ctx)
    w = np.asarray(eu.expv(1.0, op, b, m=m, ishermitian=False))
    print(opt, eu.expv.last_stats, np.linalg.norm(w - want) / np.linalg.norm(want))

"""expv through a matrix-free operator (callback) whose matvec is the library's own SpMV of the C2 operator: what the step costs
beyond the operator application itself."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
for n in (100000, 1000000):
    A = c2_operator(n)
    op = eu.MIOperator(A, ctx)
    At = torch.sparse_csr_tensor(torch.as_tensor(A.indptr, dtype=torch.int64), torch.as_tensor(A.indices, dtype=torch.int64), torch.as_tensor(A.data), size=A.shape, device="cuda")
    mf = eu.MIOperator(None, ctx, matvec=lambda x: At @ x, shape=(n, n), dtype=np.float64, ishermitian=False)
    b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    mv = lambda: At @ x
    mv(); ctx.sync(); torch.cuda.synchronize()
    t_mv = timed(mv, 50, 5, torch.cuda.synchronize)
    for name, o in (("stored operator", op), ("matrix-free (torch CSR matvec)", mf)):
        f = lambda: eu.expv(1.0, o, b, m=30, ishermitian=False, out=w)
        f(); ctx.sync()
        t = timed(f, 10, 2, ctx.sync)
        print("n=%d %-32s ms/expv %.3f  us/step %.1f  path %s   (torch matvec alone: %.1f us)" % (n, name, 1e3 * t, 1e6 * t / 30, "+".join(eu.expv.last_stats["path"]), 1e6 * t_mv))

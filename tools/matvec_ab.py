"""mul!(y, A, x) on the 2-D grid stencil: natural ordering (general diagonal form) against the grid-patch ordering (SELL slots of the stored
ordering + the permutation of x and y); device vectors."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
k = 1000; n = k * k
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
x = torch.randn(n, dtype=torch.float64, device="cuda")
for patch in (0, 1):
    ctx = eu.Context(async_outputs=True)
    ctx.set_option("patch", patch)
    op = eu.MIOperator(A, ctx)
    y = torch.empty_like(x)
    f = lambda: op.matvec(x)
    for _ in range(5): f()
    ctx.sync(); ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50): f()
        ctx.sync(); ts.append((time.perf_counter() - t0) / 50)
    print("patch=%d: mul! %.1f us" % (patch, 1e6 * sorted(ts)[2]), flush=True)

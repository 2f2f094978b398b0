"""Banded operators wider than the halo form's 8 rows (thin 2-D grids with rows of k < 64 cells: offsets -k, -1, 0, 1, k): the wave form
against the patch form in the operator's own ordering (ring = 2 k rows per tile).  EXPV_MI_RING_BAND_MAX=<rows> python tools/wide_band_ab.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader, bench
eu = expv_mi_loader.load()
n, m = 1_000_000, 30
b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
for k in (12, 24, 40, 60, 100):
    A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    ctx = eu.Context(async_outputs=True)
    op = eu.MIOperator(A, ctx)
    f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
    for _ in range(3): f()
    ctx.sync(); ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10): f()
        ctx.sync(); ts.append((time.perf_counter() - t0) / 10)
    t = sorted(ts)[2]
    print("k=%d: %.3f ms, %.3f of the contract, path %s, ring %s" % (k, 1e3 * t, bench.alg_bytes_expv(n, A.nnz, m) / t / 8e12, eu.expv.last_stats["path"], op.patch_info["longest_ring"]), flush=True)

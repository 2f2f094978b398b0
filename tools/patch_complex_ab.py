"""Complex element types on a 2-D grid (k x k): patch form (context option patch = 1) against the natural ordering (two-kernel step).
   Schroedinger: real symmetric 5-point operator + potential, complex vector, imaginary time (Lanczos); grid: complex non-Hermitian stencil
   (Arnoldi, m = 14).   python tools/patch_complex_ab.py [k] [complex64]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
k = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 1000
T = np.complex64 if "complex64" in sys.argv else np.complex128
n = k * k
rng = np.random.default_rng(1)
pot = 0.3 * rng.random(n)
S = sp.diags([np.full(n - k, 1.0), np.full(n - 1, 1.0), -4.0 + pot, np.full(n - 1, 1.0), np.full(n - k, 1.0)], [-k, -1, 0, 1, k], shape=(n, n), format="csr").astype(T)
G = (sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr") * (1 + 0.25j)).tocsr().astype(T)
b = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(T)).cuda()
s = np.dtype(T).itemsize
for name, A, t, m, herm in (("schroedinger (Lanczos, t = -0.6i, m = 30)", S, -0.6j, 30, True), ("complex grid stencil (Arnoldi, m = 14)", G, 0.7, 14, False)):
    res = {}
    for patch in (0, 1):
        ctx = eu.Context(async_outputs=True)
        ctx.set_option("patch", patch)
        op = eu.MIOperator(A, ctx)
        w = torch.empty_like(b)
        f = lambda: eu.expv(t, op, b, m=m, ishermitian=herm, out=w)
        for _ in range(3):
            f()
        ctx.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                f()
            ctx.sync()
            ts.append((time.perf_counter() - t0) / 20)
        tt = sorted(ts)[2]
        AB = A.nnz * (s + 4) + 4 * (n + 1)
        balg = (m * (AB + s * n * 4) + s * n * (m + 3)) if herm else (m * AB + s * n * (m * (m + 1) // 2 + 3 * m + 3))
        res[patch] = (tt, w.clone())
        print("%s %s patch=%d: %.3f ms (%.2f us/step), %.3f of the contract, path %s" % (name, np.dtype(T).name, patch, 1e3 * tt, 1e6 * tt / m, balg / tt / 8e12, eu.expv.last_stats["path"]), flush=True)
        del op, ctx
    print("   |w1 - w0| / |w0| = %.2e, time ratio %.3f" % (float(torch.linalg.norm(res[1][1] - res[0][1]) / torch.linalg.norm(res[0][1])), res[1][0] / res[0][0]))

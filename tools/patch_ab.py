"""2-D grid stencil, n = k^2: the patch form of the single-pass step (context option patch=1: operator stored in a grid-patch ordering,
ring recomputed) against the wave form in the natural ordering (per-tile flags).  Same inputs, results compared; the oracle check
of the patch form is in tests/test_gpu_parity.py.

   python tools/patch_ab.py [k] [float32] [serial]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
args = [a for a in sys.argv[1:]]
f32 = "float32" in args
serial = "serial" in args
ks = [int(a) for a in args if a.isdigit()] or [1000]
dt = np.float32 if f32 else np.float64
tdt = torch.float32 if f32 else torch.float64
m = 30


def alg_bytes(n, nnz, m, s):
    return m * (nnz * (s + 4) + 4 * (n + 1)) + s * n * (m * (m + 1) // 2 + 3 * m + 3)


for k in ks:
    n = k * k
    A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr").astype(dt)
    b = torch.from_numpy(np.random.default_rng(3).standard_normal(n).astype(dt)).cuda()
    res = {}
    for patch in (0, 1):
        ctx = eu.Context(async_outputs=True)
        ctx.set_option("patch", patch)
        if serial:
            ctx.set_option("pipeline_serial", 1)
        t0 = time.perf_counter()
        op = eu.MIOperator(A, ctx)
        ctx.sync()
        setup = time.perf_counter() - t0
        w = torch.empty(n, dtype=tdt, device="cuda")
        f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
        for _ in range(3):
            f()
        ctx.sync()
        path = list(eu.expv.last_stats["path"])
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                f()
            ctx.sync()
            ts.append((time.perf_counter() - t0) / 20)
        t = sorted(ts)[2]
        frac = alg_bytes(n, A.nnz, m, A.dtype.itemsize) / t / 8e12
        res[patch] = (t, w.clone(), path)
        info = dict(op.reorder_info, **(op.patch_info if patch else {}))
        print("k=%d n=%d %s patch=%d: %.3f ms per expv (%.2f us/step), %.3f of the contract; path %s; setup %.2f s; reorder %s"
              % (k, n, dt.__name__, patch, 1e3 * t, 1e6 * t / m, frac, path, setup, info), flush=True)
        del op, ctx
    d = float(torch.linalg.norm(res[1][1].double() - res[0][1].double()) / torch.linalg.norm(res[0][1].double()))
    print("k=%d: |w_patch - w_wave| / |w| = %.2e ; patch / wave time = %.3f" % (k, d, res[1][0] / res[0][0]))

"""kiops / phiv_timestep / Lanczos expv on a 2-D grid stencil: natural ordering (patch = 0) against the grid-patch ordering (patch = 1)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = k * k
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
As = sp.diags([0.5, 1.0, -3.0, 1.0, 0.5], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
rng = np.random.default_rng(1)
b = torch.from_numpy(rng.standard_normal(n)).cuda()
B = torch.from_numpy(np.asfortranarray(rng.standard_normal((n, 2)))).cuda()
def med(f, sync, reps=5, inner=5):
    f(); sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            f()
        sync()
        ts.append((time.perf_counter() - t0) / inner)
    return sorted(ts)[len(ts) // 2]
for patch in (0, 1):
    ctx = eu.Context(async_outputs=True)
    ctx.set_option("patch", patch)
    op, ops = eu.MIOperator(A, ctx), eu.MIOperator(As, ctx)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    r = {}
    r["expv m=30"] = med(lambda: eu.expv(1.0, op, b, m=30, ishermitian=False, out=w), ctx.sync)
    r["lanczos m=30"] = med(lambda: eu.expv(1.0, ops, b, m=30, ishermitian=True, out=w), ctx.sync)
    r["expv iop=2 m=30"] = med(lambda: eu.expv(1.0, op, b, m=30, iop=2, ishermitian=False, out=w), ctx.sync)
    r["kiops"] = med(lambda: eu.kiops(1.0, op, b, ishermitian=False), ctx.sync, 3, 3)
    r["kiops K=1"] = med(lambda: eu.kiops(1.0, op, B, ishermitian=False), ctx.sync, 3, 3)
    r["phiv_timestep adaptive K=1"] = med(lambda: eu.phiv_timestep(np.array([1.0]), op, B, adaptive=True, tol=1e-7), ctx.sync, 3, 3)
    r["phiv k=2 m=30"] = med(lambda: eu.phiv(1.0, op, b, 2, m=30), ctx.sync, 3, 3)
    print("patch=%d: " % patch + "; ".join("%s %.3f ms" % (a, 1e3 * t) for a, t in r.items()), flush=True)

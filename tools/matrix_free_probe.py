"""expv through a matrix-free operator (callback = the C2 stencil as six torch elementwise launches), context option matfree_fused
1 (two-kernel step fed by the callback) against 0 (modular path), and the stored operator beside it."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed, C2_OFFSETS, C2_VALS
eu = expv_mi_loader.load()
def stencil_mul(x):
    y = x * C2_VALS[2]
    for off, cv in zip(C2_OFFSETS, C2_VALS):
        if off < 0: y[-off:].add_(x[:off], alpha=cv)
        elif off > 0: y[:-off].add_(x[off:], alpha=cv)
    return y
for n in (100000, 1000000):
    A = c2_operator(n)
    b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t_mv = timed(lambda: stencil_mul(x), 50, 5, torch.cuda.synchronize)
    res = {}
    for name, fused in (("stored operator", None), ("matrix-free, two-kernel step", 1), ("matrix-free, modular path", 0)):
        ctx = eu.Context(async_outputs=True)
        if fused is not None:
            ctx.set_option("matfree_fused", fused)
            o = eu.MIOperator(None, ctx, matvec=stencil_mul, shape=(n, n), dtype=np.float64, ishermitian=False)
        else:
            o = eu.MIOperator(A, ctx)
        for kw in (dict(m=30), dict(m=30, iop=2)):
            f = lambda: eu.expv(1.0, o, b, ishermitian=False, out=w, **kw)
            f(); ctx.sync()
            t = min(timed(f, 10, 2, ctx.sync) for _ in range(3))
            res[(name, str(kw))] = w.clone()
            print("n=%d %-30s %-22s ms/expv %.3f  us/step %.1f  path %s   (callback alone: %.1f us)" % (n, name, kw, 1e3 * t, 1e6 * t / 30, "+".join(eu.expv.last_stats["path"]), 1e6 * t_mv), flush=True)
    for kw in ("{'m': 30}", "{'m': 30, 'iop': 2}"):
        ref = res[("stored operator", kw)]
        print("   rel diff to the stored operator %s: fused %.2e  modular %.2e" % (kw, float(torch.linalg.norm(res[("matrix-free, two-kernel step", kw)] - ref) / torch.linalg.norm(ref)),
              float(torch.linalg.norm(res[("matrix-free, modular path", kw)] - ref) / torch.linalg.norm(ref))))

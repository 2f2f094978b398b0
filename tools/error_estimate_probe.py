"""expv(...; mode = :error_estimate) on the symmetric C2 operator: blocks of steps through the ordinary factorisation vs the
step-by-step form (context option ee_blocked = 1 / 0), ms per call and stopping step at several rtol."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1000000
op = eu.MIOperator(c2_operator(n, sym=True), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
ref = {}
for blocked in (1, 0, 1, 0):
    ctx.set_option("ee_blocked", blocked)
    for rtol in (1e-4, 1e-8, 1e-12):
        f = lambda: eu.expv(1.0, op, b, m=30, mode="error_estimate", rtol=rtol)
        w = f(); ctx.sync()
        t = timed(f, 20, 2, ctx.sync)
        wn = np.asarray(w.cpu() if hasattr(w, "cpu") else w)
        d = 0.0 if rtol not in ref else float(np.linalg.norm(wn - ref[rtol]) / np.linalg.norm(ref[rtol]))
        ref.setdefault(rtol, wn)
        print("blocked" if blocked else "stepwise", "rtol %g" % rtol, "steps", eu.expv.last_subspace.m, "ms %.3f" % (1e3 * t), "rel diff to first %.1e" % d)
plain = lambda: eu.expv(1.0, op, b, m=30, ishermitian=True)
plain(); ctx.sync()
print("plain Lanczos expv, m = 30: ms %.3f" % (1e3 * timed(plain, 20, 2, ctx.sync)))

import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1000000
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for name, f in (("expv", lambda: eu.expv(1.0, op, b, m=30, ishermitian=False)),
                ("arnoldi", lambda: eu.arnoldi(op, b, m=30, ishermitian=False)),
                ("phiv k=1", lambda: eu.phiv(1.0, op, b, 1, m=30, ishermitian=False)),
                ("phiv k=4", lambda: eu.phiv(1.0, op, b, 4, m=30, ishermitian=False)),
                ("phiv k=4 correct", lambda: eu.phiv(1.0, op, b, 4, m=30, ishermitian=False, correct=True))):
    f(); ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(5): f()
    ctx.sync(); prof = ctx.prof_get(); ctx.prof_enable(False)
    t = timed(f, 20, 2, ctx.sync)
    print("%-18s ms %.3f   kernels us/call: %s" % (name, 1e3 * t, {k: round(1e3 * v["total_ms"] / 5, 1) for k, v in prof.items()}))

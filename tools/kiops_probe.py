"""kiops on the real C2 operator: wall time per call against the spans of its device work."""
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
f = lambda: eu.kiops(1.0, op, b, ishermitian=False, opnorm=4.4)
w, st = f(); ctx.sync()
print("stats", st)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5): f()
ctx.sync(); prof = ctx.prof_get(); ctx.prof_enable(False)
t = timed(f, 20, 2, ctx.sync)
print("ms per call %.3f   device spans per call (us): %s" % (1e3 * t, {k: (v["launches"] // 5, round(1e3 * v["total_ms"] / 5, 1)) for k, v in prof.items()}))

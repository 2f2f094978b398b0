"""Per-call latency distribution of the integrator-facing calls (n = 1e6, C2 operator): are there multi-millisecond outliers?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1000000
op = eu.MIOperator(c2_operator(n), ctx)
ops = eu.MIOperator(c2_operator(n, sym=True), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
calls = {
    "expv": lambda: eu.expv(1.0, op, b, m=30, ishermitian=False),
    "expv_timestep adaptive": lambda: eu.expv_timestep([0.5, 1.0], op, b, tol=1e-6, adaptive=True),
    "error_estimate": lambda: eu.expv(1.0, ops, b, m=30, mode="error_estimate", rtol=1e-8),
}
for name, f in calls.items():
    for _ in range(5): f()
    ctx.sync()
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); f(); ctx.sync(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    print("%-24s median %.3f ms  p90 %.3f  p99 %.3f  max %.3f  calls > 3 ms: %d of %d" % (name, np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), int((ts > 3).sum()), len(ts)), flush=True)
    if (ts > 3).any():
        print("   outliers at", np.nonzero(ts > 3)[0].tolist()[:20], [round(x, 1) for x in ts[ts > 3][:10]])
print(ctx.counters())

"""expv on small systems (the bench's small_systems entry), for A/B runs of two builds: EXPV_MI_LIB=<other .so> python tools/small_ab.py
   [sym]   -- sym: the symmetric operator (Lanczos)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader, bench
eu = expv_mi_loader.load()
sym = len(sys.argv) > 1 and sys.argv[1] == "sym"
ctx = eu.Context(async_outputs=True)
out = {}
for n in (20_000, 60_000, 100_000, 130_000, 200_000, 1_000_000):
    op = eu.MIOperator(bench.c2_operator(n, sym=sym), ctx)
    b = torch.randn(n, dtype=torch.float64, device="cuda")
    w = torch.empty_like(b)
    f = lambda: eu.expv(1.0, op, b, m=30, ishermitian=sym, out=w)
    for _ in range(5): f()
    ctx.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40): f()
        ctx.sync()
        ts.append((time.perf_counter() - t0) / 40)
    t = sorted(ts)[2]
    out[n] = (1e3 * t, 1e6 * t / 30)
print(("lanczos " if sym else "arnoldi ") + "  ".join("n=%d: %.3f ms (%.2f us/step)" % (n, a, b) for n, (a, b) in out.items()))

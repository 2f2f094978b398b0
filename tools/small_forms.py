"""Small systems: ms per expv (C2 operator, m = 30) by step form -- overlapped launches, the resident form (one cooperative kernel), one launch
after the other.
    python tools/small_forms.py [n ...]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
sizes = [int(float(a)) for a in sys.argv[1:]] or [20000, 50000, 100000, 200000, 500000]
ctx = eu.Context(async_outputs=True)
for n in sizes:
    op = eu.MIOperator(c2_operator(n), ctx)
    b = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.empty_like(b)
    row = {"n": n}
    ref = None
    for name, opts in (("overlapped", {}), ("resident", {"resident": 1}), ("serial", {"pipeline_serial": 1})):
        for k, v in {"resident": 0, "pipeline_serial": 0, **opts}.items():
            ctx.set_option(k, v)
        for _ in range(5):
            eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
        ctx.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(100):
                eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 100)
        if ref is None:
            ref = w.clone()
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(10):
            eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
        ctx.sync()
        prof = ctx.prof_get(); ctx.prof_enable(False)
        row[name] = {"ms": round(1e3 * best, 4), "us_per_step": round(1e6 * best / 30, 2), "path": "+".join(eu.expv.last_stats["path"]),
                     "diff": float((w - ref).abs().max()),
                     "kernels_us_per_expv": {k: round(1e3 * v["total_ms"] / 10, 1) for k, v in prof.items()}}
    print(row)
    del op

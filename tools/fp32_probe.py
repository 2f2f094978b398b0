import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader, bench
eu = expv_mi_loader.load()
n, m = 1_000_000, 30
ctx = eu.Context(async_outputs=True)
A = bench.c2_operator(n)
op64 = eu.MIOperator(A, ctx)
op32 = eu.MIOperator(A.astype(np.float32), ctx)
b = torch.as_tensor(np.random.default_rng(3).standard_normal(n), device="cuda")
b32 = b.to(torch.float32)
w = torch.empty(n, dtype=torch.float64, device="cuda"); w32 = torch.empty(n, dtype=torch.float32, device="cuda")
for name, f in (("fp64", lambda: eu.expv(1.0, op64, b, m=m, ishermitian=False, out=w)), ("fp32", lambda: eu.expv(1.0, op32, b32, m=m, ishermitian=False, out=w32))):
    for mode in ("overlapped", "serial"):
        ctx.set_pipeline_overlap(mode == "overlapped")
        f(); ctx.sync()
        c0 = ctx.counters()
        t = bench.timed(f, 20, 3, ctx.sync)
        c1 = ctx.counters()
        print(name, mode, "ms %.4f" % (1e3 * t), eu.expv.last_stats["path"], {k: c1[k] - c0[k] for k in c0})
        if mode == "serial":
            ctx.prof_reset(); ctx.prof_enable(True)
            for _ in range(5): f()
            ctx.sync(); pr = ctx.prof_get(); ctx.prof_enable(False)
            print("   kernels", {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in pr.items()})
ctx.set_pipeline_overlap(True)
torch.cuda.synchronize(); ctx.sync()
print("rel diff fp32 vs fp64:", float(torch.linalg.norm(w32.double() - w) / torch.linalg.norm(w)), float(w32.abs().sum()), float(w.abs().sum()))

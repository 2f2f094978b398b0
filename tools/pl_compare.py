"""expv(1.0, A, b; m = 30) on the symmetric C2 operator by n: default Lanczos path vs ortho = "pipelined" (stream-ordered outputs, ms per call)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
for n in (20000, 50000, 100000, 200000, 500000, 1000000, 2000000, 4000000):
    A = c2_operator(n, sym=True)
    ctx = eu.Context(async_outputs=True)
    op = eu.MIOperator(A, ctx)
    bt = torch.randn(n, dtype=torch.float64, device="cuda")
    wt = torch.empty_like(bt)
    r = {}
    for ortho in ("auto", "pipelined", "auto", "pipelined"):
        for _ in range(5):
            eu.expv(1.0, op, bt, m=30, ishermitian=True, ortho=ortho, out=wt)
        ctx.sync()
        reps = 40 if n <= 200000 else 15
        t0 = time.perf_counter()
        for _ in range(reps):
            eu.expv(1.0, op, bt, m=30, ishermitian=True, ortho=ortho, out=wt)
        ctx.sync()
        r.setdefault(ortho, []).append(1e3 * (time.perf_counter() - t0) / reps)
    print("n=%8d   default %.3f / %.3f ms   pipelined %.3f / %.3f ms   ratio %.2f" % (n, r["auto"][0], r["auto"][1], r["pipelined"][0], r["pipelined"][1],
                                                                                     min(r["pipelined"]) / min(r["auto"])), flush=True)

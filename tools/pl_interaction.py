"""does a pipelined-Lanczos call disturb the default path afterwards?  default x 20, pipelined x 20, default x 20 ... on one context, by n"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
for n in (100000, 500000, 100000):
    A = c2_operator(n, sym=True)
    ctx = eu.Context(async_outputs=True)
    op = eu.MIOperator(A, ctx)
    bt = torch.randn(n, dtype=torch.float64, device="cuda")
    wt = torch.empty_like(bt)
    out = []
    for ortho in ("auto", "auto", "pipelined", "auto", "auto", "pipelined", "auto"):
        c0 = ctx.counters()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            eu.expv(1.0, op, bt, m=30, ishermitian=True, ortho=ortho, out=wt)
        ctx.sync()
        c1 = ctx.counters()
        out.append("%s %.3f (redo %d)" % (ortho[:4], 1e3 * (time.perf_counter() - t0) / 20, c1["redo_serial"] - c0["redo_serial"]))
    print(n, " | ".join(out), flush=True)

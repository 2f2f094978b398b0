"""Cost of a Krylov step as a function of the window length: expv at m = 4, 8, 12, ... on the C2 pattern (fp64 and x (1 + 0.25i)
ComplexF64); the difference between two runs / the steps between them = us per step of that window range, beside the contract's
bytes for those steps.  usage: python tools/window_cost.py [n] [serial]      (serial: launches one after the other, option pipeline_serial)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed, a_bytes
eu = expv_mi_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = eu.Context(async_outputs=True)
if len(sys.argv) > 2 and sys.argv[2] == "serial":
    ctx.set_option("pipeline_serial", 1)
    print("(serial mode)")
for name, A, s in (("fp64", c2_operator(n), 8), ("complex", (c2_operator(n) * (1 + 0.25j)).tocsr(), 16)):
    op = eu.MIOperator(A, ctx)
    rng = np.random.default_rng(3)
    b = torch.as_tensor(rng.standard_normal(n) + (1j * rng.standard_normal(n) if s == 16 else 0.0), device="cuda")
    w = torch.empty_like(b)
    prev_t, prev_m = None, 0
    for m in (2, 4, 8, 12, 16, 20, 24, 28, 30, 31):
        f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
        f(); ctx.sync()
        t = min(timed(f, 20, 2, ctx.sync) for _ in range(3))
        line = "%-8s m=%2d  %8.1f us/expv  path %s" % (name, m, 1e6 * t, "+".join(eu.expv.last_stats["path"]))
        if prev_t is not None:
            steps = m - prev_m
            us = 1e6 * (t - prev_t) / steps
            AB = a_bytes(n, A.nnz, s)
            byts = sum(AB + s * n * (j + 2) for j in range(prev_m + 1, m + 1)) / steps + s * n      # + the combine's extra column per step of m
            line += "   steps %2d..%2d: %6.1f us/step, contract %.0f MB/step -> %.2f of 8 TB/s" % (prev_m + 1, m, us, byts / 1e6, byts / (us * 1e-6) / 8e12)
        print(line, flush=True)
        prev_t, prev_m = t, m

"""pipelined Lanczos: time of lanczos! alone by Krylov dimension (fixed cost of a call vs cost per pass), default path beside it"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
for n in (20000, 100000, 1000000, 4000000):
    A = c2_operator(n, sym=True)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    bt = torch.randn(n, dtype=torch.float64, device="cuda")
    res = {}
    for ortho in ("pipelined", "auto"):
        for m in (2, 16, 30):
            Ks = eu.KrylovSubspace(np.float64, np.float64, n, 30, 0, ctx)
            for _ in range(3):
                eu.lanczos_(Ks, op, bt, m=m, ortho=ortho)
                _ = Ks.m
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                eu.lanczos_(Ks, op, bt, m=m, ortho=ortho)
                _ = Ks.m
            ctx.sync()
            res[(ortho, m)] = 1e6 * (time.perf_counter() - t0) / 20
    for ortho in ("pipelined", "auto"):
        r = res
        print("n=%d %-9s lanczos!: m=2 %.0f us, m=16 %.0f us, m=30 %.0f us  -> per pass %.1f us (16..30), fixed ~%.0f us" % (
            n, ortho, r[(ortho, 2)], r[(ortho, 16)], r[(ortho, 30)], (r[(ortho, 30)] - r[(ortho, 16)]) / 14, r[(ortho, 2)] - 2 * (r[(ortho, 30)] - r[(ortho, 16)]) / 14), flush=True)

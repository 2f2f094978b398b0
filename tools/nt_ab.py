"""Non-temporal window loads by footprint: where is the crossover?  expv on the C2 pattern (fp64 and ComplexF64) at several n and m with the
context option nontemporal = 0 (never), 1 (always), -1 (by footprint: the product).  Three contexts, measured interleaved; note that two contexts in
the SAME mode can differ by 15 % at n = 1e6 (placement of their basis in memory / the Infinity Cache), so read differences below that with care.  usage: python tools/nt_ab.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
for name, cx, n in (("complex", True, 1_000_000), ("fp64", False, 1_000_000), ("fp64", False, 1_500_000), ("fp64", False, 2_000_000), ("complex", True, 700_000)):
    A = c2_operator(n)
    if cx:
        A = (A * (1 + 0.25j)).tocsr()
    rng = np.random.default_rng(3)
    b = torch.as_tensor(rng.standard_normal(n) + (1j * rng.standard_normal(n) if cx else 0.0), device="cuda")
    w = torch.empty_like(b)
    res = {}
    ctxs = {}
    for mode in (-1, 0, 1):
        ctx = eu.Context(async_outputs=True)
        ctx.set_option("nontemporal", mode)
        ctxs[mode] = (ctx, eu.MIOperator(A, ctx))
    for m in (16, 20, 24, 30):
        for rnd in range(3):                      # interleaved: clocks and caches see the three modes alike
            for mode in (-1, 0, 1):
                ctx, op = ctxs[mode]
                f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
                f(); ctx.sync()
                t = 1e6 * timed(f, 10, 2, ctx.sync)
                res[(mode, m)] = min(res.get((mode, m), 1e30), t)
    ctxs.clear()
    s = 16 if cx else 8
    print("%-8s n=%8d (column %5.1f MB)  " % (name, n, s * n / 1e6) + "   ".join("m=%d: never %7.1f always %7.1f product %7.1f us" % (m, res[(0, m)], res[(1, m)], res[(-1, m)]) for m in (16, 20, 24, 30)), flush=True)

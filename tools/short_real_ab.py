"""Real short windows, n = 1e6 (and 1e5): Hermitian Lanczos, iop = 2, kiops on the C2 pattern, us per call -- A/Bs of the reduction chain (EXPV_MI_LIB)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
for n in (1_000_000, 100_000):
    rng = np.random.default_rng(6)
    A = c2_operator(n)
    Ah = sp.diags([0.5, 1.0, -3.0, 1.0, 0.5], [-2, -1, 0, 1, 2], shape=(n, n), format="csr")
    b = torch.as_tensor(rng.standard_normal(n), device="cuda"); w = torch.empty_like(b)
    op, oph = eu.MIOperator(A, ctx), eu.MIOperator(Ah, ctx)
    cases = (("lanczos m=30", lambda: eu.expv(1.0, oph, b, m=30, ishermitian=True, out=w)),
             ("iop=2 m=30", lambda: eu.expv(1.0, op, b, m=30, iop=2, ishermitian=False, out=w)),
             ("full m=8", lambda: eu.expv(1.0, op, b, m=8, ishermitian=False, out=w)),
             ("full m=30", lambda: eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)),
             ("kiops real", lambda: eu.kiops(1.0, op, b, ishermitian=False, opnorm=4.6)))
    for name, f in cases:
        f(); ctx.sync()
        t = min(timed(f, 20, 2, ctx.sync) for _ in range(5))
        print("n=%-8d %-14s %8.1f us per call" % (n, name, 1e6 * t), flush=True)

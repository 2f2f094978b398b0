"""Complex short windows (Hermitian Lanczos, iop = 2, kiops on the complex C2 pattern), n = 1e6: us per call, best of five blocks.
A/B of pipe.hip knobs that touch the short complex variants: run once per library (EXPV_MI_LIB).  usage: python tools/short_complex_ab.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1_000_000
rng = np.random.default_rng(6)
Ac = (c2_operator(n) * (1 + 0.25j)).tocsr()
Ah = sp.diags([0.3 - 0.2j, -2.0, 0.3 + 0.2j], [-1, 0, 1], shape=(n, n), format="csr")        # Hermitian: Lanczos, windows of 2 columns
b = torch.as_tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n), device="cuda")
w = torch.empty_like(b)
opc, oph = eu.MIOperator(Ac, ctx), eu.MIOperator(Ah, ctx)
cases = (("lanczos complex m=30", lambda: eu.expv(1.0, oph, b, m=30, ishermitian=True, out=w)),
         ("iop=2 complex m=30", lambda: eu.expv(1.0, opc, b, m=30, iop=2, ishermitian=False, out=w)),
         ("full complex m=8", lambda: eu.expv(1.0, opc, b, m=8, ishermitian=False, out=w)),
         ("kiops complex (C4)", lambda: eu.kiops(1.0, opc, b, allow_complex=True, ishermitian=False, opnorm=4.6)))
for name, f in cases:
    f(); ctx.sync()
    t = min(timed(f, 20, 2, ctx.sync) for _ in range(5))
    print("%-24s %8.1f us per call" % (name, 1e6 * t), flush=True)

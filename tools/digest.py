"""Digest of a fixed set of results (bit-for-bit comparison of two builds: EXPV_MI_LIB=<other .so> python tools/digest.py)."""
import hashlib, sys
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
torch.manual_seed(0)
rng = np.random.default_rng(0)
def h(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
out = {}
for n in (1000, 20000, 300000, 1000000):
    A = c2_operator(n)
    b = torch.randn(n, dtype=torch.float64, device="cuda")
    op = eu.MIOperator(A)
    out["expv_%d" % n] = h(eu.expv(1.0, op, b, m=30, ishermitian=False))
    out["expv_m12_%d" % n] = h(eu.expv(0.5, op, b, m=12, ishermitian=False))
    S = (A + A.T) * 0.5
    out["lanczos_%d" % n] = h(eu.expv(1.0, eu.MIOperator(S.tocsc()), b, m=30, ishermitian=True))
    out["phiv_%d" % n] = h(eu.phiv(0.7, op, b, 3, m=20))
n = 50000
A = sp.random(n, n, density=8.0 / n, random_state=1, format="csc") - 2.0 * sp.eye(n, format="csc")
b = torch.randn(n, dtype=torch.float64, device="cuda")
out["sell_expv"] = h(eu.expv(0.3, eu.MIOperator(A.tocsc()), b, m=25, ishermitian=False))
Ac = (c2_operator(40000) * (1.0 + 0.3j)).tocsc()
bc = torch.randn(40000, dtype=torch.complex128, device="cuda")
out["cplx_expv"] = h(eu.expv(1.0, eu.MIOperator(Ac), bc, m=14, ishermitian=False))
D = torch.randn(1500, 1500, dtype=torch.float64, device="cuda") / 40.0
bd = torch.randn(1500, dtype=torch.float64, device="cuda")
out["dense_expv"] = h(eu.expv(1.0, eu.MIOperator(D), bd, m=30, ishermitian=False))
n = 100000
op = eu.MIOperator(c2_operator(n))
B = torch.randn(n, 3, dtype=torch.float64, device="cuda")
out["kiops"] = h(eu.kiops(1.0, op, B, tol=1e-8)[0])
ts = [0.1, 0.5, 1.0]
out["timestep"] = h(eu.expv_timestep(ts, op, B[:, 0].contiguous(), tol=1e-8))
for k in sorted(out):
    print(k, out[k])

"""What the convenience forms cost on top of the device-resident call (n = 1e6, C2 operator, m = 30): host vectors (numpy) in and
out, and a scipy matrix passed directly (content fingerprint per call)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
ctx = eu.Context()
n = 1000000
A = c2_operator(n)
op = eu.MIOperator(A, ctx)
bh = np.random.default_rng(0).standard_normal(n)
bd = torch.as_tensor(bh, device="cuda")
for name, f in (("device b, device result", lambda: eu.expv(1.0, op, bd, m=30, ishermitian=False)),
                ("host b (numpy), host result", lambda: eu.expv(1.0, op, bh, m=30, ishermitian=False)),
                ("host b, host result, scipy matrix passed directly", lambda: eu.expv(1.0, A, bh, m=30, ishermitian=False))):
    f(); ctx.sync()
    print("%-52s ms %.3f" % (name, 1e3 * timed(f, 20, 2, ctx.sync)))

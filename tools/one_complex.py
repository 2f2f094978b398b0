"""A few expv calls on the C2 pattern x (1 + 0.25i), ComplexF64, m = 30 (bench key c2_complex_full_arnoldi) -- the command profiled by
tools/prof_r05.sh for the HBM traffic of the 16 / 24 / 32-column complex kernels."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator((c2_operator(n) * (1 + 0.25j)).tocsr(), ctx)
rng = np.random.default_rng(6)
b = torch.as_tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n), device="cuda")
w = torch.empty_like(b)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
print("path", eu.expv.last_stats["path"])

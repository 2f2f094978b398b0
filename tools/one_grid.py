"""a few whole-call expv on the 2-D grid stencil (k x k, offsets -k,-1,0,1,k), for kernel traces:
   python tools/one_grid.py [k] [calls] [patch 0|1] [float32]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
patch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f32 = len(sys.argv) > 4 and sys.argv[4] == "float32"
n = k * k
ctx = eu.Context(async_outputs=True)
ctx.set_option("patch", patch)
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
if f32:
    A = A.astype(np.float32)
op = eu.MIOperator(A, ctx)
b = torch.randn(n, dtype=torch.float32 if f32 else torch.float64, device="cuda")
w = torch.empty_like(b)
for _ in range(calls):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
print(float(w.abs().sum()), eu.expv.last_stats["path"])

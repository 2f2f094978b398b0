import sys, os
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import expv_mi_loader
eu = expv_mi_loader.load()
from oracle import krylov_oracle as ko
rng = np.random.default_rng(29)
k = 640; n = k * k
G = sp.diags([0.7, 1.1, -4.0, 0.9, 1.3], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
q = rng.permutation(n)
A = G[q][:, q].tocsr().astype(np.float32)
b = rng.standard_normal(n).astype(np.float32)
A64, b64 = A.astype(np.float64), b.astype(np.float64)
ctx = eu.Context(); ctx.set_option("patch", 0)
op = eu.MIOperator(A, ctx)
for m, iop in ((12, 0), (20, 3), (6, 0), (9, 0), (16, 0)):
    wo = ko.expv(0.3, A64, b64, m=m, iop=iop, ishermitian=False)
    for rep in range(3):
        ctx.set_pipeline_overlap(True)
        w = np.asarray(eu.expv(0.3, op, b, m=m, iop=iop, ishermitian=False)).copy(); p1 = list(eu.expv.last_stats["path"])
        ctx.set_pipeline_overlap(False)
        w2 = np.asarray(eu.expv(0.3, op, b, m=m, iop=iop, ishermitian=False)).copy()
        e = lambda x: float(np.linalg.norm(x.astype(np.float64) - wo) / np.linalg.norm(wo))
        print("m=%d iop=%d rep %d: overlapped vs oracle %.2e  serial vs oracle %.2e  equal %s  path %s" % (m, iop, rep, e(w), e(w2), np.array_equal(w, w2), p1), flush=True)
print("---- V of 6 steps, overlapped against serial")
ctx.set_pipeline_overlap(True)
K1 = eu.arnoldi(op, b, m=6, ishermitian=False); V1 = np.asarray(K1.getV()).copy(); H1 = np.asarray(K1.getH()).copy()
ctx.set_pipeline_overlap(False)
K2 = eu.arnoldi(op, b, m=6, ishermitian=False); V2 = np.asarray(K2.getV()).copy(); H2 = np.asarray(K2.getH()).copy()
print("H diff", np.abs(H1 - H2).max())
for c in range(V1.shape[1]):
    d = np.nonzero(V1[:, c] != V2[:, c])[0]
    if len(d):
        t = d // 1024
        print("column %d: %d rows differ, first rows %s, tiles %s ... (%d tiles), lanes-in-tile(row%%1024//4) %s" % (c, len(d), d[:8], np.unique(t)[:12], len(np.unique(t)), np.unique((d % 1024) // 4)[:16]))
    else:
        print("column %d equal" % c)

"""Creation cost of the three reordered bench operators (shuffled band / grid / triangulated mesh, n = 1e6), phases printed by the
library under EXPV_MI_OP_TIMING=1; second creation of the same pattern = the plan cache.  usage: python tools/op_create_unstructured.py"""
import os, sys, time
os.environ["EXPV_MI_OP_TIMING"] = "1"
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context()
n, kq = 1_000_000, 1000
def band():
    q = np.random.default_rng(17).permutation(n)
    return c2_operator(n)[q][:, q].tocsr()
def grid():
    q = np.random.default_rng(18).permutation(n)
    return sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-kq, -1, 0, 1, kq], shape=(n, n), format="csr")[q][:, q].tocsr()
def mesh():
    rg = np.random.default_rng(19); ii = np.arange(n); parts = []
    for dr, dc in ((0, 0), (0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, -1)):
        r, c = ii // kq + dr, ii % kq + dc
        ok = (r >= 0) & (r < kq) & (c >= 0) & (c < kq)
        v = (-2.0 if (dr, dc) == (0, 0) else 0.4) + 0.05 * rg.random(n)
        parts.append(sp.csr_matrix((v[ok], (ii[ok], (r * kq + c)[ok])), shape=(n, n)))
    q = rg.permutation(n)
    return sum(parts).tocsr()[q][:, q].tocsr()
for name, mk in (("shuffled band", band), ("shuffled grid", grid), ("shuffled mesh", mesh)):
    A = mk()
    for rep in (1, 2):
        sys.stderr.flush()
        print("==== %s, creation %d" % (name, rep), flush=True)
        t = time.perf_counter()
        op = eu.MIOperator(A, ctx)
        ctx.sync()
        print("==== %s, creation %d: %.3f s  %s" % (name, rep, time.perf_counter() - t, op.reorder_info), flush=True)
        del op

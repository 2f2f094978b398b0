"""A/B of the two-kernel step's first kernel (context option fa2_pipelined): ms per expv(1.0, A, b; m = 30) and bitwise equality of the
results on the bench's non-local operators -- uniformly random columns (fp64), power-law rows, local columns, and the reference's GPU-test
operator shape (ComplexF64 sprand).   python tools/fa2_ab.py [reps]"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np, torch, scipy.sparse as sp
import expv_mi_loader
from bench import general_sparse_operator
eu = expv_mi_loader.load()
n, m = 1_000_000, 30
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = eu.Context(async_outputs=True)


def gputest_operator():
    rgp = np.random.default_rng(41)
    nz1, nz2 = 10 * n, n
    r1, c1_ = rgp.integers(0, n, nz1), rgp.integers(0, n, nz1)
    keep = c1_ > r1
    A = (sp.csr_matrix(((rgp.random(nz1) + 1j * rgp.random(nz1))[keep], (r1[keep], c1_[keep])), shape=(n, n))
         + sp.csr_matrix((rgp.random(nz2) + 1j * rgp.random(nz2), (rgp.integers(0, n, nz2), rgp.integers(0, n, nz2))), shape=(n, n))).tocsr()
    A.sum_duplicates()
    return A


cases = [("random", lambda: general_sparse_operator("random", n), np.float64, 1.0),
         ("powerlaw", lambda: general_sparse_operator("powerlaw", n), np.float64, 1.0),
         ("local", lambda: general_sparse_operator("local", n), np.float64, 1.0),
         ("sprand_complex", gputest_operator, np.complex128, 0.1)]
for name, mk, dt, t in cases:
    A = mk()
    op = eu.MIOperator(A, ctx)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    if dt == np.complex128:
        b = torch.complex(b, torch.randn(n, dtype=torch.float64, device="cuda", generator=g))
    res = {}
    for opt in (0, 1, 0, 1):
        ctx.set_option("fa2_pipelined", opt)
        w = torch.empty_like(b)
        eu.expv(t, op, b, m=m, ishermitian=False, out=w)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            eu.expv(t, op, b, m=m, ishermitian=False, out=w)
        ctx.sync()
        dtm = (time.perf_counter() - t0) / reps
        res.setdefault(opt, []).append((dtm, w.clone()))
    path = eu.expv.last_stats["path"]
    same = bool(torch.equal(res[0][0][1], res[1][0][1]))
    print("%-16s path %-28s plain %.3f / %.3f ms   pipelined %.3f / %.3f ms   ratio %.3f   bitwise equal: %s" % (
        name, path, 1e3 * res[0][0][0], 1e3 * res[0][1][0], 1e3 * res[1][0][0], 1e3 * res[1][1][0],
        min(r[0] for r in res[1]) / min(r[0] for r in res[0]), same), flush=True)
    del op, A

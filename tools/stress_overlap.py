"""Race hunt for the overlapped pipeline: many factorisations of random sizes, each run with consecutive steps on two
streams and again one launch after the other; both must agree to the last bit (same kernels, same arithmetic).
usage: python tools/stress_overlap.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import expv_mi_loader
from tests._util import c2_operator

eu = expv_mi_loader.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(1234)
ctx = eu.Context()
ops = {}
t0 = time.time()
calls = bad = 0
while time.time() - t0 < budget:
    n = int(rng.choice([300, 513, 2048, 5000, 20011, 65536, 150000, 400000]))
    m = int(rng.integers(1, 33))
    iop = int(rng.choice([0, 0, 0, 2, 5]))
    kind = int(rng.integers(0, 2))          # 0: narrow band (halo form), 1: structured-grid offsets (wave form)
    if (n, kind) not in ops:
        if kind == 0:
            ops[(n, kind)] = eu.MIOperator(c2_operator(n), ctx)
        else:
            import scipy.sparse as sp
            k = max(3, int(np.sqrt(n)) // 2)
            ops[(n, kind)] = eu.MIOperator(sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr"), ctx)
    b = rng.standard_normal(n)
    res = []
    for overlap in (True, False):
        ctx.set_pipeline_overlap(overlap)
        for rep in range(3 if overlap else 1):        # back-to-back overlapped calls reuse flags, mailbox, workspace
            w = eu.expv(float(rng.choice([0.3, 1.0])) if False else 0.7, ops[(n, kind)], b, m=m, iop=iop, ishermitian=False)
            if overlap:
                res.append(np.asarray(w).copy())
        if not overlap:
            ref = np.asarray(w).copy()
    ctx.set_pipeline_overlap(True)
    calls += 4
    for r in res:
        if not np.array_equal(r, ref):
            bad += 1
            print("MISMATCH n=%d kind=%d m=%d iop=%d maxdiff=%g" % (n, kind, m, iop, float(np.max(np.abs(r - ref)))), flush=True)
print("calls %d, mismatches %d, %.0f s" % (calls, bad, time.time() - t0))
sys.exit(1 if bad else 0)

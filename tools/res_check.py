"""Resident form (context option "resident" = 1, pipe.hip k_pipe_resident) against the step-wise single-pass form on BASELINE
config 2: time per expv and the difference of the results.  usage: python tools/res_check.py [n]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
import expv_mi_loader
eu = expv_mi_loader.load()
n, m = (int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000), 30
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csc")
op = eu.MIOperator(A)
b = torch.randn(n, dtype=torch.float64, device="cuda")
w0 = None
for res in (0, 1, 1):
    eu.default_context().set_option("resident", res)
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        w = eu.expv(1.0, op, b, m=m, ishermitian=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print("resident", res, "ms/expv %.4f" % (dt * 1e3), "path", eu.expv.last_stats["path"])
    if res == 0:
        w0 = w.clone()
    else:
        print("  rel diff vs step-wise: %.3e" % float((w - w0).norm() / w0.norm()))

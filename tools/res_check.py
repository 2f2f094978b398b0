import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, scipy.sparse as sp, torch
import expv_mi_loader
eu = expv_mi_loader.load()
n, m = 1_000_000, 30
rng = np.random.default_rng(0)
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csc")
op = eu.MIOperator(A)
b = torch.randn(n, dtype=torch.float64, device="cuda")
ctx = eu.default_context() if hasattr(eu, "default_context") else None
def run(res):
    os.environ.pop("X", None)
    c = eu.get_context() if hasattr(eu, "get_context") else None
    return None
import time
for res in (0, 1, 1):
    try:
        eu.default_context().set_option("resident", res)
    except Exception as e:
        print("set_option failed", e)
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w = eu.expv(1.0, op, b, m=m, ishermitian=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("resident", res, "ms/expv %.4f" % (dt * 1e3), "path", eu.expv.last_stats.get("path") if hasattr(eu.expv, "last_stats") else None, "norm", float(w.norm()))
    if res == 0: w0 = w.clone()
    else: print("  rel diff vs stepwise: %.3e" % float((w - w0).norm() / w0.norm()))

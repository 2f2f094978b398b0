"""One problem with as many rows as the 64-problem sub-chunk of config 5 has in total (n = 6.4e6, m = 30): does the single-pass
step stream as fast on a 1.6 GB basis as on the 248 MB one of config 2?  (serial form: per-kernel durations are meaningful)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 6_400_000, 30
ctx = eu.Context()
ctx.set_option("pipeline_serial", 1)
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(2):
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
tot = sum(5 * n * 8 + n * 8 * (j + 2) for j in range(1, m + 1))
print({"n": n, "ms_per_call": 1e3 * dt, "alg_GB": tot / 1e9, "TBps_whole_call": tot / dt / 1e12, "path": eu.expv.last_stats["path"]})

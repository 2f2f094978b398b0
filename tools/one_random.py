"""A few expv calls on the bench's uniformly-random-column operator (general_sparse_random: two-kernel step), for a kernel trace:
rocprofv3 --kernel-trace ... -- python tools/one_random.py [calls] [kind]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import general_sparse_operator
eu = expv_mi_loader.load()
n = 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
ctx = eu.Context(async_outputs=True)
if len(sys.argv) > 3:
    ctx.set_option("fa2_pipelined", int(sys.argv[3]))      # round 6: A/B of the two-kernel step's first kernel
op = eu.MIOperator(general_sparse_operator(kind, n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
w = torch.empty_like(b)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
print("path", eu.expv.last_stats["path"])

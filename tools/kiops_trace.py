"""a few kiops calls on the real C2 operator for a kernel trace: rocprofv3 --kernel-trace ... -- python tools/kiops_trace.py"""
import sys
sys.path.insert(0, ".")
import torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
ctx = eu.Context(async_outputs=True)
n = 1000000
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(6):
    w, st = eu.kiops(1.0, op, b, ishermitian=False, opnorm=4.4)
ctx.sync()
print(st)

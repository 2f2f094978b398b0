"""a few whole-call expv on the config-2 operator (device result, stream-ordered), for kernel traces: python tools/one_expv.py [n] [calls]"""
import sys
sys.path.insert(0, ".")
import torch
import expv_mi_loader, bench
eu = expv_mi_loader.load()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(bench.c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
w = torch.empty_like(b)
for _ in range(calls):
    eu.expv(1.0, op, b, m=30, ishermitian=False, out=w)
ctx.sync()
print(float(w.abs().sum()))

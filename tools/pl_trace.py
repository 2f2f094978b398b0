"""per-pass stamps of sampled workgroups of the pipelined Lanczos kernel (build: tools/build_variant.py pltrace -DPL_TRACE on lanczos_pl.hip)"""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
import exponentialutilities_jl_amd._lib as L
lib = L.load()
n, m = 1000000, 30
A = c2_operator(n, sym=True)
ctx = eu.Context()
op = eu.MIOperator(A, ctx)
bt = torch.randn(n, dtype=torch.float64, device="cuda")
Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
for _ in range(3):
    eu.lanczos_(Ks, op, bt, m=m, ortho="pipelined"); _ = Ks.m
buf = (C.c_ulonglong * (8 * 40 * 6))()
raw = C.CDLL(L.LIB_PATH)
rc = raw.expv_mi_pl_trace(buf)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 40, 6).astype(np.float64) / 100.0   # wall_clock64: 100 MHz -> us
for g in (0, 3, 7):
    print("workgroup %d:" % (g * 64 + 7))
    for k in range(10, 16):
        s = a[g, k]
        print("  pass %2d: scalars %.2f  neighbours %.2f  edge tiles + ack %.2f  interior %.2f  publish %.2f   | pass total %.2f us" % (
            k, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], a[g, k + 1, 0] - s[0]))

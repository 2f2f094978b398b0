"""Per-tile phases of the wave form of the single-pass step (trace build: python exponentialutilities.jl_amd/build.py --trace;
EXPV_MI_LIB=exponentialutilities.jl_amd/libexpv_mi_trace.so python tools/wave_trace.py KIND out.txt [opt=val ...])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import expv_mi_loader
eu = expv_mi_loader.load()
from exponentialutilities_jl_amd import _lib as L
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("gs", os.path.join(os.path.dirname(os.path.abspath(__file__)), "general_sparse.py"))
gs = importlib.util.module_from_spec(spec); spec.loader.exec_module(gs)
kind, out = sys.argv[1], sys.argv[2]
n, m = 1_000_000, 30
ctx = eu.Context(async_outputs=True)
for o in sys.argv[3:]:
    ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
A = gs.make(kind, n)
op = eu.MIOperator(A, ctx)
b = torch.as_tensor(np.random.default_rng(3).standard_normal(n), device="cuda")
for _ in range(6):
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
ctx.sync()
print("path", eu.expv.last_stats["path"])
lib = L.load()
lib.expv_mi_pipe_trace_dump.argtypes = [ctypes.c_char_p]
lib.expv_mi_pipe_trace_dump(out.encode())
d = np.loadtxt(out + ".wave", dtype=np.int64)
st, blk, tl = d[:, 0], d[:, 1], d[:, 2]
T = d[:, 3:] * 0.01      # us (100 MHz wall clock)
print("step: tiles/WG | per tile medians (us): loads+update | store ack+flag | wait neighbours | gather+y | sums | tile total || kernel span (first start -> last end)")
for q in sorted(set(st)):
    mk = st == q
    t = T[mk]
    ph = np.diff(t, axis=1)
    ok = (t > 0).all(axis=1)
    ph = ph[ok]
    span = t[ok][:, 5].max() - t[ok][:, 0].min()
    print(q, int(tl[mk].max()) + 1, "|", " | ".join("%.2f" % np.median(ph[:, k]) for k in range(5)), "| %.2f" % np.median(t[ok][:, 5] - t[ok][:, 0]), "|| %.1f" % span)

"""Per-step / per-workgroup timeline of the banded pipeline (config 2).  Needs the trace build:
  hipcc ... -DPIPE_TRACE -c csrc/pipe.hip -o build/pipe_trace.o ; link with the other objects into
  libexpv_mi_trace.so ; EXPV_MI_LIB=<that .so> python tools/pipe_trace.py gpurun_out/pipe_trace.txt"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import expv_mi_loader
eu = expv_mi_loader.load()
from exponentialutilities_jl_amd import _lib as L
n, m = int(float(os.environ.get("TRACE_N", "1e6"))), 30
offs = [-1000, -1, 0, 1, 1000] if (len(sys.argv) > 2 and sys.argv[2] == "stencil") else [-2, -1, 0, 1, 2]
sym = os.environ.get("TRACE_SYM", "0") == "1"      # symmetric operator: the Lanczos variant (window 2)
A = sp.diags([0.5, 1.0, -3.0, 1.0, 0.5] if sym else [0.3, 1.2, -2.0, 0.8, -0.1], offs, shape=(n, n), format="csc")
op = eu.MIOperator(A)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(12):
    w = eu.expv(1.0, op, b, m=m, ishermitian=sym)
torch.cuda.synchronize()
lib = L.load()
lib.expv_mi_pipe_trace_dump.argtypes = [ctypes.c_char_p]
lib.expv_mi_pipe_trace_dump(sys.argv[1].encode())
d = np.loadtxt(sys.argv[1], dtype=np.int64)
step, blk, t0, t1, t2, t3, t4, t5 = d.T[:8]
fine = d.T[8:] if d.shape[1] > 8 else None
prev_pub = None
print("per step, us relative to the previous step's publish (step 1: to its own first start):")
print("step nblk | kernel start first/median/last | released (after wait) median/last | main end median/last | reduced | epilogue stored | published  [= step time]")
for q in sorted(set(step)):
    mk = step == q
    ref = prev_pub if prev_pub is not None else t0[mk].min()
    r = lambda x: round((x - ref) * 0.01, 1)
    rel = t4[mk][t4[mk] > 0]
    pub = t3[mk].max()
    red = t2[mk].max()
    epi = t5[mk].max()
    print(q, mk.sum(), "|", r(t0[mk].min()), r(np.median(t0[mk])), r(t0[mk].max()), "|",
          (float(r(np.median(rel))), float(r(rel.max()))) if len(rel) else "-", "|", r(np.median(t1[mk])), r(t1[mk].max()), "|", r(red), "|",
          (r(epi) if epi > 0 else "-"), "|", r(pub))
    prev_pub = pub

if fine is not None and len(fine) >= 5:
    t6, t7, t8, t9, t10 = fine[:5]
    print("first tile of a workgroup, medians in us: released -> loads landed (coefficients in LDS) -> halo done -> u_j in LDS -> y~ stored -> [sums, other tiles] main end -> partials published")
    for q in sorted(set(step)):
        mk = (step == q) & (t4 > 0) & (t6 > 0)
        if not mk.any(): continue
        md = lambda a, b: round(float(np.median((a[mk] - b[mk]) * 0.01)), 2)
        print(q, md(t6, t4), md(t7, t6), md(t8, t7), md(t9, t8), md(t1, t9), md(t10, t1))

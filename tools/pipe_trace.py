"""Per-step / per-workgroup timeline of the banded pipeline (config 2).  Needs the trace build:
  hipcc ... -DPIPE_TRACE -c csrc/pipe.hip -o build/pipe_trace.o ; link with the other objects into
  libexpv_mi_trace.so ; EXPV_MI_LIB=<that .so> python tools/pipe_trace.py gpurun_out/pipe_trace.txt"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import expv_mi_loader
eu = expv_mi_loader.load()
from exponentialutilities_jl_amd import _lib as L
n, m = 1_000_000, 30
A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csc")
op = eu.MIOperator(A)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(12):
    w = eu.expv(1.0, op, b, m=m, ishermitian=False)
torch.cuda.synchronize()
lib = L.load()
lib.expv_mi_pipe_trace_dump.argtypes = [ctypes.c_char_p]
lib.expv_mi_pipe_trace_dump(sys.argv[1].encode())
d = np.loadtxt(sys.argv[1], dtype=np.int64)
step, blk, t0, t1, t2, t3, t4, t5 = d.T
prev_pub = None
print("step nblk | pass begin: first/median/last (us after previous publish) | main end median/last (us after first begin) | reduced | published")
for q in sorted(set(step)):
    mk = step == q
    b0 = t0[mk].min()
    last = t2[mk] > 0
    red = (t2[mk][last].max() - b0) * 0.01 if last.any() else -1
    pub = (t3[mk].max() - b0) * 0.01 if (t3[mk] > 0).any() else -1
    rel = (lambda x: (x - prev_pub) * 0.01) if prev_pub is not None else (lambda x: (x - b0) * 0.01)
    gs = t4[mk][t4[mk] > 0]
    grp = f"group stage done: first {((gs.min() - b0) * 0.01):.1f} last {((gs.max() - b0) * 0.01):.1f}; blk0 main end {((t1[mk][blk[mk] == 0][0] - b0) * 0.01):.1f} blk0 group done {((t4[mk][blk[mk] == 0][0] - b0) * 0.01):.1f}" if len(gs) else ""
    print(q, mk.sum(), "|", round(rel(t0[mk].min()), 1), round(rel(np.median(t0[mk])), 1), round(rel(t0[mk].max()), 1), "|",
          round((np.median(t1[mk]) - b0) * 0.01, 1), round((t1[mk].max() - b0) * 0.01, 1), "|", round(red, 1), "|", round(pub, 1), "|", grp)
    prev_pub = t3[mk].max() if (t3[mk] > 0).any() else None

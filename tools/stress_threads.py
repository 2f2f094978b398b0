"""Two (or more) host threads, each with its own context, calling the library at the same time (ctypes releases the GIL inside the
calls): results against the same calls made alone.    python tools/stress_threads.py [seconds] [threads]"""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np
import expv_mi_loader
from tests._util import c2_operator
eu = expv_mi_loader.load()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sizes = [3000, 65536, 400000]
refs = {}
ctx0 = eu.Context()
bs = {n: np.random.default_rng(n).standard_normal(n) for n in sizes}
for n in sizes:
    for sym in (False, True):
        op = eu.MIOperator(c2_operator(n, sym=sym), ctx0)
        refs[(n, sym)] = np.asarray(eu.expv(0.7, op, bs[n], m=25, ishermitian=sym)).copy()
out = []
def work(tid):
    ctx = eu.Context()
    ops = {(n, sym): eu.MIOperator(c2_operator(n, sym=sym), ctx) for n in sizes for sym in (False, True)}
    rng = np.random.default_rng(tid)
    calls = bad = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        n = int(rng.choice(sizes)); sym = bool(rng.integers(0, 2))
        w = np.asarray(eu.expv(0.7, ops[(n, sym)], bs[n], m=25, ishermitian=sym))
        calls += 1
        if not np.array_equal(w, refs[(n, sym)]):
            rel = float(np.linalg.norm(w - refs[(n, sym)]) / np.linalg.norm(refs[(n, sym)]))
            if rel > 1e-12:
                bad += 1
    out.append({"thread": tid, "calls": calls, "wrong": bad, "counters": ctx.counters()})
ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
[t.start() for t in ts]; [t.join() for t in ts]
for o in out: print(o)
sys.exit(1 if any(o["wrong"] for o in out) else 0)

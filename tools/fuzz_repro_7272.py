"""seed 7272 case 5412 of tests/fuzz_parity.py (round 6 soak): adaptive phiv_timestep with caches -- the device's controller ended in
'1000 proposals' where the harness expected a result.  Prints the case, both controllers' outcomes and their last log lines."""
import sys
sys.path.insert(0, ".")
import numpy as np
import tests.fuzz_parity as fz
rec = {}
saved = (fz.eu.phiv_timestep_, fz.ko.phiv_timestep_)
def cap_dev(*a, **kw):
    rec.setdefault("dev", []).append((a, dict(kw)))
    return saved[0](*a, **kw)
def cap_ref(*a, **kw):
    rec.setdefault("ref", []).append((a, dict(kw)))
    return saved[1](*a, **kw)
fz.eu.phiv_timestep_ = cap_dev
fz.ko.phiv_timestep_ = cap_ref
try:
    print(fz.one_case(7272, 5412))
except Exception as e:
    print("harness:", repr(e)[:300])
fz.eu.phiv_timestep_, fz.ko.phiv_timestep_ = saved
for side in ("dev", "ref"):
    for a, kw in rec.get(side, []):
        U, ts, A, B = a[:4]
        print(side, "T", getattr(A, "dtype", None), "shape", getattr(A, "shape", None), "ts", np.asarray(ts), "B", np.asarray(B).shape, np.asarray(B).dtype, {k: v for k, v in kw.items() if k != "caches"})
# run both with a log
(a, kw) = rec["dev"][-1]
log = []
try:
    fz.eu.phiv_timestep_(np.empty_like(a[0]), np.asarray(a[1]).copy(), a[2], a[3], **{**{k: v for k, v in kw.items() if k != "caches"}, "verbose": True, "out": log.append})
    print("device: finished;", len(log), "log lines")
except Exception as e:
    print("device:", repr(e)[:200], "|", len(log), "log lines; last:", log[-3:])
if True:
    (a, kw) = rec["ref"][-1] if "ref" in rec else rec["dev"][-1]
    st = {}
    try:
        olog = []
        A_o = a[2].astype(np.complex128) if hasattr(a[2], "astype") else a[2]
        fz.ko.phiv_timestep_(np.empty(np.asarray(a[0]).shape, dtype=np.complex128), np.asarray(a[1]).copy(), A_o, np.asarray(a[3]).astype(np.complex128),
                             **{**{k: v for k, v in kw.items() if k != "caches"}, "stats": st, "verbose": True, "out": olog.append})
        print("oracle: finished", st, "| last log:", olog[-3:])
    except Exception as e:
        print("oracle:", repr(e)[:200], "| last log:", olog[-3:] if olog else None)

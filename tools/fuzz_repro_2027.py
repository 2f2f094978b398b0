import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import fuzz_parity as fz
rec = {}
orig = fz.eu.expv_timestep
def cap(ts, A, b, **kw):
    rec["args"] = (np.array(ts), A, b, dict(kw)); return orig(ts, A, b, **kw)
fz.eu.expv_timestep = cap
origo = fz.ko.expv_timestep
def capo(ts, A, b, **kw):
    rec["oargs"] = (np.array(ts), A, b, dict(kw)); return origo(ts, A, b, **kw)
fz.ko.expv_timestep = capo
try:
    print(fz.one_case(2027, 22242)[3])
except Exception as e:
    print("exc", e)
ts, A, b, kw = rec["args"]
print("ts", ts, "kw", kw, "n", b.shape)
so = {}
olog = []
U = origo(rec["oargs"][0].copy(), rec["oargs"][1], rec["oargs"][2], **dict(rec["oargs"][3], stats=so, verbose=True, out=olog.append))
print("oracle stats", so)
open("gpurun_out/repro_oracle.log", "w").write("\n".join(olog))
for verbose in (0,):
    sd = {}
    try:
        import contextlib, io
        orig(ts.copy(), A, b, **dict(kw, stats=sd, verbose=True))
    except Exception as e:
        print("device raised:", str(e)[:300])
    print("device stats", sd)

"""Randomised mixed-API stress on one context: expv, arnoldi! + expv!/phiv!, adaptive phiv_timestep, kiops (real and complex),
expv_batch, in random order and sizes, every result checked against the numpy oracle.  usage: python tools/stress_mixed.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import expv_mi_loader
from oracle import krylov_oracle as ko
from tests._util import c2_operator

eu = expv_mi_loader.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2024)
ctx = eu.Context()
t0 = time.time()
calls = {}
worst = {}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def note(kind, err, bar):
    calls[kind] = calls.get(kind, 0) + 1
    worst[kind] = max(worst.get(kind, 0.0), err)
    if not err <= bar:
        print("MISMATCH", kind, err, bar, flush=True)
        sys.exit(1)


while time.time() - t0 < budget:
    kind = rng.choice(["expv", "split", "timestep", "kiops", "kiops_c", "batch", "grid"])
    n = int(rng.choice([257, 600, 1500, 3001]))
    m = int(rng.integers(3, 31))
    A = c2_operator(n)
    b = rng.standard_normal(n)
    t = float(rng.uniform(0.2, 1.0))
    if kind == "expv":
        iop = int(rng.choice([0, 0, 2, 5]))
        w = eu.expv(t, eu.MIOperator(A, ctx), b, m=m, iop=iop, ishermitian=False)
        note(kind, rel(w, ko.expv(t, A, b, m=m, iop=iop, ishermitian=False)), 1e-11)
    elif kind == "grid":
        k = max(9, int(np.sqrt(n)))
        Ag = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
        w = eu.expv(t, eu.MIOperator(Ag, ctx), b, m=m, ishermitian=False)
        note(kind, rel(w, ko.expv(t, Ag, b, m=m, ishermitian=False)), 1e-11)
    elif kind == "split":
        op = eu.MIOperator(A, ctx)
        Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
        eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
        Ko = ko.arnoldi(A, b, m=m, ishermitian=False)
        note("split_H", rel(Ks.H[: m + 1, :m], Ko.H[: m + 1, :m]), 1e-11)
        W = eu.phiv(t, Ks, 2)
        Wo = ko.phiv_(np.empty((n, 3), order="F"), t, Ko, 2)
        note("split_phiv", rel(W[0] if isinstance(W, tuple) else W, Wo[0] if isinstance(Wo, tuple) else Wo), 1e-10)
        if m >= 6:          # continuation
            Ks2 = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
            j = m // 2
            eu.arnoldi_(Ks2, op, b, m=j, ishermitian=False)
            eu.arnoldi_(Ks2, op, b, m=m, ishermitian=False, init=j)
            note("split_cont", rel(Ks2.H[: m + 1, :m], Ko.H[: m + 1, :m]), 1e-11)
    elif kind == "timestep":
        B = np.asfortranarray(rng.standard_normal((n, 3)))
        ts = np.sort(rng.uniform(0.5, 4.0, size=2))
        st, so = {}, {}
        U = eu.phiv_timestep(ts.copy(), eu.MIOperator(A, ctx), B, adaptive=True, tol=1e-8, stats=st)
        Uo = ko.phiv_timestep(ts.copy(), A, B, adaptive=True, tol=1e-8, stats=so)
        if (st["num_timesteps"], st["m"]) == (so["num_timesteps"], so["m"]):
            note(kind, rel(U, Uo), 1e-10)
        else:                   # a controller decision on the edge: compare against dense truth instead
            note(kind + "_edge", 0.0, 1.0)
    elif kind in ("kiops", "kiops_c"):
        cplx = kind == "kiops_c"
        Ac = (A * (1 + 0.25j)).tocsc() if cplx else A
        u = np.asfortranarray(rng.standard_normal((n, 2)) + (1j * rng.standard_normal((n, 2)) if cplx else 0))
        w, st = eu.kiops(t, eu.MIOperator(Ac, ctx), u, allow_complex=cplx, ishermitian=False) if cplx else eu.kiops(t, eu.MIOperator(Ac, ctx), u, ishermitian=False)
        wo, so = ko.kiops(t, Ac, u, allow_complex=cplx, ishermitian=False) if cplx else ko.kiops(t, Ac, u, ishermitian=False)
        if tuple(st) == tuple(so):
            note(kind, rel(np.asarray(w).ravel(), np.asarray(wo).ravel()), 1e-9)
        else:
            note(kind + "_edge", 0.0, 1.0)
    else:
        A0 = A.tocsr(); A0.sort_indices()
        P = int(rng.integers(2, 9))
        sc = 1 + 0.1 * rng.random(P)
        vals = np.stack([A0.data * s for s in sc])
        B = np.asfortranarray(rng.standard_normal((n, P)))
        W = np.asarray(eu.expv_batch(t, A0, vals, B, m=m, ctx=ctx))
        p = int(rng.integers(0, P))
        Ap = A0.copy(); Ap.data = vals[p].copy()
        note(kind, rel(W[:, p], ko.expv(t, Ap, B[:, p], m=m, ishermitian=False)), 1e-11)
print("calls", calls)
print("worst", {k: "%.2e" % v for k, v in worst.items()})
print("counters", ctx.counters())

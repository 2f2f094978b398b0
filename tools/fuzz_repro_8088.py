import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
import tests.fuzz_parity as fz
rec = {}
orig = fz.eu.expv_batch
def cap(*a, **kw):
    rec["a"] = (a, dict(kw)); return orig(*a, **kw)
fz.eu.expv_batch = cap
print(fz.one_case(8088, 4782)[1:3])
(t, P, vals, Bm), kw = rec["a"]
print("t", t, "n", P.shape, "nprob", vals.shape, kw, "|vals| max", np.abs(vals).max())
W = np.asarray(orig(t, P, vals, Bm, **kw))
eu, ko = fz.eu, fz.ko
for q in range(vals.shape[0]):
    Aq = P.copy(); Aq.data = vals[q].copy()
    A64 = Aq.astype(np.complex128); b = Bm[:, q].astype(np.complex128)
    wo = ko.expv(t, A64, b, m=kw["m"], iop=kw["iop"], ishermitian=False)
    ws = np.asarray(eu.expv(t, Aq, b, m=kw["m"], iop=kw["iop"], ishermitian=False))
    Ko = ko.arnoldi(A64, b, m=kw["m"], iop=kw["iop"], ishermitian=False)
    Vo = Ko.getV()[:, :Ko.m]
    loss = np.max(np.abs(Vo.conj().T @ Vo - np.eye(Ko.m)))
    Ks = eu.arnoldi(Aq, b, m=kw["m"], iop=kw["iop"], ishermitian=False)
    Vd = np.asarray(Ks.getV())[:, :Ks.m]
    lossd = np.max(np.abs(Vd.conj().T @ Vd - np.eye(Ks.m)))
    Hd, Ho = np.asarray(Ks.getH()), Ko.getH()
    truth = None
    r = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    print("problem %d: batch vs oracle %.2e  single vs oracle %.2e  batch vs single %.2e | oracle orth loss %.2e device orth loss %.2e | H err %.2e  |H| %.2e  H[m+1,m] %.2e %.2e" % (
        q, r(W[:, q], wo), r(ws, wo), r(W[:, q], ws), loss, lossd, np.max(np.abs(Hd - Ho)) / np.max(np.abs(Ho)), np.max(np.abs(Ho)), abs(Ho[Ko.m, Ko.m - 1]), abs(Hd[Ks.m, Ks.m - 1])))
    print("   subdiagonal of H (oracle):", np.array2string(np.abs(np.diag(Ho, -1)), precision=2))

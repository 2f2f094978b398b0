"""Pipelined Lanczos (ortho = "pipelined") against the oracle's reference recurrence and the numpy restatement of the device scheme."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import expv_mi_loader
from oracle import krylov_oracle as ko
from oracle import pipelined_lanczos as pl
from tests._util import c2_operator
eu = expv_mi_loader.load()
rng = np.random.default_rng(5)
for n, m in ((2000, 30), (70001, 30), (513, 7), (100000, 30), (1000000, 30)):
    A = c2_operator(n, sym=True)
    b = rng.standard_normal(n)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
    eu.lanczos_(Ks, op, b, m=m, ortho="pipelined")
    H = np.asarray(Ks.getH()).copy()
    w = np.asarray(eu.expv_(np.empty(n), 0.7, Ks))
    Ko = ko.KrylovSubspace(np.float64, np.float64, n, m)
    ko.lanczos_(Ko, A, b, m=m)
    wo = ko.expv_(np.empty(n), 0.7, Ko)
    Ho = Ko.getH()
    b0, al, be, V3 = pl.lanczos_p3(A, b, m)
    k = Ks.m
    V = np.asarray(Ks.getV())
    print("n=%d m=%d Ks.m=%d breakdown=%s  |w-w_ref|/|w| = %.2e  max|H-H_ref|/max|H| = %.2e  vs numpy p3: alpha %.2e beta %.2e  |V'V-I| = %.2e  beta %.3e vs %.3e" % (
        n, m, k, Ks.wasbreakdown, np.linalg.norm(w - wo) / np.linalg.norm(wo), np.abs(H[:k + 1, :k] - Ho[:k + 1, :k]).max() / np.abs(Ho).max(),
        np.abs(np.diag(H)[:k] - al[:k]).max(), np.abs(np.diag(H, -1)[:k] - be[:k]).max(), np.abs(V[:, :k].T @ V[:, :k] - np.eye(k)).max(), Ks.beta, Ko.beta), flush=True)
# timing at n = 1e6
n, m = 1000000, 30
import torch
A = c2_operator(n, sym=True)
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(A, ctx)
bt = torch.randn(n, dtype=torch.float64, device="cuda")
wt = torch.empty_like(bt)
for ortho in ("auto", "pipelined", "auto", "pipelined"):
    for _ in range(3):
        eu.expv(1.0, op, bt, m=m, ishermitian=True, ortho=ortho, out=wt)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        eu.expv(1.0, op, bt, m=m, ishermitian=True, ortho=ortho, out=wt)
    ctx.sync()
    print(ortho, "%.3f ms per expv" % (1e3 * (time.perf_counter() - t0) / 20), eu.expv.last_stats["path"])

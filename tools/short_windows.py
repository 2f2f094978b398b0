"""The short-window entries of the bench line in one quick run: Lanczos (C2 symmetric), kiops on the real C2 operator, kiops on the
complex one (BASELINE configs[3]), complex Hermitian Lanczos; ms per call, us per Krylov step, result digests (bit-exactness across
A/B builds).  usage: [EXPV_MI_PIPE_PF=0] [EXPV_MI_LIB=...] python tools/short_windows.py [n]"""
import sys, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator, timed
eu = expv_mi_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = eu.Context(async_outputs=True)
rng = np.random.default_rng(3)
b = torch.as_tensor(rng.standard_normal(n), device="cuda"); w = torch.empty_like(b)
bc = torch.as_tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n), device="cuda"); wc = torch.empty_like(bc)
dig = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
def best(f, reps=20):
    f(); ctx.sync()
    return min(timed(f, reps, 2, ctx.sync) for _ in range(5))
ops = eu.MIOperator(c2_operator(n, sym=True), ctx)
t = best(lambda: eu.expv(1.0, ops, b, m=30, ishermitian=True, out=w))
print("lanczos            %8.1f us/call  %6.2f us/step  digest %s" % (1e6 * t, 1e6 * t / 30, dig(w)), flush=True)
Hc = (c2_operator(n, sym=True) * 1.0).astype(np.complex128)
Hc = Hc + 0.3j * (np.abs(Hc - Hc.T) * 0)        # Hermitian (real symmetric values as ComplexF64)
opc_h = eu.MIOperator(Hc.tocsr(), ctx)
t = best(lambda: eu.expv(-0.6j, opc_h, bc, m=30, ishermitian=True, out=wc))
print("lanczos complex    %8.1f us/call  %6.2f us/step  digest %s" % (1e6 * t, 1e6 * t / 30, dig(wc)), flush=True)
op = eu.MIOperator(c2_operator(n), ctx)
st = {}
def kr():
    st["w"], st["s"] = eu.kiops(1.0, op, b, ishermitian=False, opnorm=4.4)
c0 = ctx.counters(); kr(); ctx.sync(); c1 = ctx.counters()
steps = c1["krylov_steps"] - c0["krylov_steps"]
t = best(kr, 10)
print("kiops real         %8.1f us/call  %6.2f us/step  (%d steps) stats %s digest %s" % (1e6 * t, 1e6 * t / steps, steps, tuple(st["s"]), dig(torch.as_tensor(st["w"]))), flush=True)
opc = eu.MIOperator((c2_operator(n) * (1 + 0.25j)).tocsr(), ctx)
def kc():
    st["w"], st["s"] = eu.kiops(1.0, opc, bc, allow_complex=True, ishermitian=False, opnorm=4.6)
c0 = ctx.counters(); kc(); ctx.sync(); c1 = ctx.counters()
steps = c1["krylov_steps"] - c0["krylov_steps"]
t = best(kc, 10)
print("kiops complex (C4) %8.1f us/call  %6.2f us/step  (%d steps) stats %s digest %s" % (1e6 * t, 1e6 * t / steps, steps, tuple(st["s"]), dig(torch.as_tensor(st["w"]))), flush=True)
for iop in (2, 3):
    t = best(lambda: eu.expv(1.0, op, b, m=30, iop=iop, ishermitian=False, out=w))
    print("arnoldi iop=%d      %8.1f us/call  %6.2f us/step  digest %s" % (iop, 1e6 * t, 1e6 * t / 30, dig(w)), flush=True)
print("counters", {k: v for k, v in ctx.counters().items() if "redo" in k})

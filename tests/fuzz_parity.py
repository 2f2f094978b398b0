"""Randomised parity hunt against the oracle: random sizes, element types, operator structures, calls and options.
Every case is derived from (seed, index), printed on failure and reproducible with  python tests/fuzz_parity.py 0 SEED INDEX.
    python tests/fuzz_parity.py SECONDS [SEED]
Test infrastructure (imports the oracle); not part of the product or of the measured path."""
import json
import sys
import time
import traceback

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg  # noqa: F401  (sp.linalg.norm)

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import expv_mi_loader
from oracle import krylov_oracle as ko

eu = expv_mi_loader.load()


FOCUS = __import__("os").environ.get("FUZZ_FOCUS", "")      # "complex_windows": complex element types, banded / grid / wide-band operators, m in 16..40 (round 5's new kernels)


def make_operator(rng, n, cplx):
    kind = rng.choice(["banded", "banded", "wide_diagonals", "regular_rows", "irregular_rows", "dense", "symmetric_banded", "hermitian_dense", "grid2d", "wide_band"])
    if FOCUS == "complex_windows":
        kind = rng.choice(["banded", "banded", "grid2d", "wide_band", "symmetric_banded"])
    def vals(shape, scale):
        v = rng.standard_normal(shape) * scale
        return v + 1j * rng.standard_normal(shape) * scale if cplx else v
    if kind in ("banded", "symmetric_banded"):
        w = int(rng.integers(0, min(9, n)))
        offs = sorted(set([0] + [int(o) for o in rng.integers(-w, w + 1, size=int(rng.integers(1, 9)))]))
        if kind == "symmetric_banded":
            offs = sorted(set(offs + [-o for o in offs]))
        d = [vals(n - abs(o), 0.3 / np.sqrt(len(offs))) if rng.random() < 0.7 else np.full(n - abs(o), vals((), 0.3)) for o in offs]
        A = sp.diags(d, offs, shape=(n, n), format="csr")
        if kind == "symmetric_banded":
            A = ((A + A.conj().T) * 0.5).tocsr()
        A = A - 0.5 * sp.identity(n, format="csr")
    elif kind == "grid2d":
        # 5- / 9-point stencil on a 2-D grid (rows of k >= 64 cells, >= 8 grid rows, the last one possibly incomplete), with or without
        # entries across the row ends: the real element types are stored in the grid-patch ordering (patch form of the step)
        k = int(rng.integers(64, 200))
        n = k * int(rng.integers(8, 70)) + (int(rng.integers(0, k)) if rng.random() < 0.4 else 0)
        offs = [-k, -1, 0, 1, k] if rng.random() < 0.7 else [-k - 1, -k, -k + 1, -1, 0, 1, k - 1, k, k + 1]
        if rng.random() < 0.3:
            offs = [o for o in offs if o != offs[0]] or offs          # one-sided coupling
        i = np.arange(n)
        wrap = rng.random() < 0.5
        parts = []
        for o in offs:
            ok = (i + o >= 0) & (i + o < n)
            if not wrap and abs(o) <= 2:
                ok &= ((i + o) // k) == (i // k)
            v = vals(n, 0.3 / np.sqrt(len(offs)))
            parts.append(sp.csr_matrix((v[ok], (i[ok], i[ok] + o)), shape=(n, n)))
        A = (sum(parts) - 0.5 * sp.identity(n, format="csr")).tocsr()
        if rng.random() < 0.3:
            A = ((A + A.conj().T) * 0.5).tocsr()
            kind = "grid2d_symmetric"
        if rng.random() < 0.35:      # the same mesh in a random numbering: creation cuts it into patches from breadth-first distances (n >= 8192)
            q = rng.permutation(n)
            A = A[q][:, q].tocsr()
            A.sort_indices()
    elif kind == "wide_band":
        # a band of 9 .. 140 rows (thin grids, block-banded systems): up to an eighth of a tile the patch form in the operator's own ordering,
        # beyond it the wave form / mesh patches / the two-kernel step
        w = int(rng.integers(9, 141))
        n = max(n, 3 * w + 5)
        offs = sorted(set([0, -w if rng.random() < 0.8 else w] + [int(o) for o in rng.integers(-w, w + 1, size=int(rng.integers(1, 10)))]))
        d = [vals(n - abs(o), 0.3 / np.sqrt(len(offs))) for o in offs]
        A = sp.diags(d, offs, shape=(n, n), format="csr") - 0.5 * sp.identity(n, format="csr")
        if rng.random() < 0.25:
            A = ((A + A.conj().T) * 0.5).tocsr()
            kind = "wide_band_symmetric"
    elif kind == "wide_diagonals":
        nd = int(rng.integers(2, 7))
        offs = sorted(set([0] + [int(o) for o in rng.integers(-(n - 1), n, size=nd)]))
        d = [vals(n - abs(o), 0.3 / np.sqrt(len(offs))) for o in offs]
        A = sp.diags(d, offs, shape=(n, n), format="csr") - 0.5 * sp.identity(n, format="csr")
    elif kind in ("regular_rows", "irregular_rows"):
        if kind == "regular_rows":
            ln = np.full(n, int(rng.integers(1, 7)))
        else:
            ln = np.minimum(rng.zipf(1.7, size=n), max(1, min(n // 2, 3000)))
        rows = np.repeat(np.arange(n), ln)
        cols = rng.integers(0, n, size=rows.size)
        v = vals(rows.size, 0.3) / np.sqrt(np.repeat(ln, ln))
        A = (sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr() - 0.5 * sp.identity(n, format="csr")).tocsr()
        A.sum_duplicates()
    else:
        n = min(n, 400)
        A = vals((n, n), 1.0 / np.sqrt(n)) - 0.5 * np.eye(n)
        if kind == "hermitian_dense":
            A = (A + A.conj().T) * 0.5
    return kind, n, A


class _RefSpins(Exception):
    """the oracle (= the reference's controller) did not terminate within its time limit: 'reference spins'"""


def _limited(f, seconds):
    import signal

    def on_alarm(sig, frm):
        raise _RefSpins("reference spins")
    old = signal.signal(signal.SIGALRM, on_alarm)
    left = signal.alarm(seconds)
    try:
        return f()
    finally:
        signal.alarm(max(left - seconds, 1) if left else 0)
        signal.signal(signal.SIGALRM, old)


_ACTX = []


def _async_ctx():
    if not _ACTX:
        _ACTX.append(eu.Context(async_outputs=True))
    return _ACTX[0]


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def one_case(seed, index, verbose=False):
    rng = np.random.default_rng([seed, index])
    T = np.dtype(rng.choice(["float64", "float64", "complex128", "float32", "complex64"]))
    if FOCUS == "complex_windows":
        T = np.dtype(rng.choice(["complex128", "complex128", "complex64"]))
    cplx = T.kind == "c"
    T64 = np.dtype(np.complex128 if cplx else np.float64)
    single = T.itemsize == (8 if cplx else 4)
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 127, 128, 129, 511, 512, 513, 1000, 1537, 2500, 4099]))
    if rng.random() < float(__import__('os').environ.get('FUZZ_LARGE', '0.04')):            # grids of many workgroups: wave form, several tiles per workgroup, overlapped launches
        n = int(rng.choice([20000, 65537, 150001, 300000]))
    kind, n, A64 = make_operator(rng, n, cplx)
    if rng.random() < 0.25:           # other magnitudes of the operator: scaling / squaring counts, Pade degrees, slow or no convergence
        # (32-bit: |tau A| of 20-80 with a truncated orthogonalisation amplifies fp32 rounding by ~|A| / H[j+1, j] per step -- the
        #  fp64 oracle is no yardstick there, seed 102 cases 189 / 2268 -- so the large factors are for the 64-bit types)
        A64 = A64 * float(rng.choice([1e-3, 0.1, 10.0, 40.0] if T.itemsize == (16 if cplx else 8) else [1e-3, 0.1, 2.0]))
    A = A64.astype(T)
    A64 = A.astype(T64)               # the oracle sees exactly the values the device has
    b = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
    if rng.random() < 0.02:
        b[:] = 0                      # zero starting vector (arnoldi.jl:366)
    elif rng.random() < 0.05:
        b *= T.type(1e-6) if rng.random() < 0.5 else T.type(1e5)
    b64 = b.astype(T64)
    tq = float(rng.choice([0.7, 0.7, 0.7, -0.4, 1e-8, 0.0, 3.0]))      # the time of the plain calls
    m = int(rng.integers(1, 41))
    iop = int(rng.choice([0, 0, 0, 1, 2, 3, 7]))
    if FOCUS == "complex_windows":
        m = int(rng.integers(16, 41))
        iop = int(rng.choice([0, 0, 0, 17, 24, 31]))
    herm = kind in ("symmetric_banded", "hermitian_dense", "grid2d_symmetric", "wide_band_symmetric") and bool(rng.integers(0, 2))
    call = rng.choice(["expv", "expv", "arnoldi", "phiv", "expv_timestep", "phiv_timestep", "expv_complex_t", "kiops", "error_estimate",
                       "subspace_reuse", "continuation", "update_values", "matrix_free", "batch", "phiv_correct", "async_device", "caches"])
    ortho = str(rng.choice(["lowsync", "mgs"]))
    if n > 5000 and call not in ("expv", "arnoldi", "expv_complex_t", "phiv", "subspace_reuse", "update_values"):
        call = "expv"                 # (large cases: the calls whose oracle stays cheap)
    desc = {"seed": seed, "index": index, "T": T.name, "n": n, "operator": kind, "m": m, "iop": iop, "hermitian": herm, "call": str(call), "ortho": ortho}
    if verbose:
        print(desc, flush=True)
    tol = 3e-4 if single else 1e-10
    bnorm = float(np.linalg.norm(b64))
    kw = dict(m=m, iop=iop, ishermitian=herm)
    # how the caller hands the operands over (the values stay the same): CSC / CSR / COO, C- or Fortran-ordered dense, a strided view of b,
    # device-resident b
    pres = str(rng.choice(["plain", "plain", "csc", "coo", "fortran", "strided_b", "device_b", "device_dense"]))
    desc["presentation"] = pres
    Ain, bin_ = A, b
    if pres == "csc" and sp.issparse(A):
        Ain = A.tocsc()
    elif pres == "coo" and sp.issparse(A):
        Ain = A.tocoo()
    elif pres == "fortran" and not sp.issparse(A):
        Ain = np.asfortranarray(A)
    elif pres == "strided_b":
        big = np.zeros(2 * n, dtype=T)
        big[::2] = b
        bin_ = big[::2]
    elif pres == "device_b":
        import torch
        bin_ = torch.as_tensor(b, device="cuda")
    elif pres == "device_dense" and not sp.issparse(A):
        import torch                  # a device-resident dense operator, row-major (torch's default) or column-major
        Ad = torch.as_tensor(A, device="cuda")
        Ain = Ad if rng.random() < 0.5 else Ad.t().contiguous().t()
    as64 = lambda x: np.asarray(x).astype(T64 if np.asarray(x).dtype.kind == "c" or cplx else np.float64)
    def rel(a, r):
        a, r = np.asarray(a), np.asarray(r)
        if single and np.isfinite(r).all() and float(np.max(np.abs(r), initial=0.0)) > 1e37:
            return 0.0                # (beyond Float32: Inf is the right answer there)
        if not np.isfinite(a).all():
            return float("inf")
        # (a result that decayed far below ||b|| -- exp(-34) b -- is what cancellation leaves of terms of size ||b||: judged against that)
        nr = max(float(np.linalg.norm(r)), 1e-6 * bnorm)
        if not np.isfinite(np.asarray(r)).all():
            return 0.0 if not np.isfinite(a).all() else float("nan")
        return float(np.linalg.norm(a.astype(np.complex128) - r)) / (nr if nr > 0 else 1.0)
    err, extra = 0.0, {}
    if not b.any() and call not in ("expv", "expv_complex_t", "arnoldi"):
        # zero starting vector: arnoldi! returns at iszero(beta) (arnoldi.jl:366) with V `undef`; what phiv / the time steppers make of
        # that (0 x undef, x / 0.0) is not defined behaviour -- expv and arnoldi themselves are compared (zero result, m = 0 rules)
        call = desc["call"] = "expv"
    if call == "expv":
        w = eu.expv(tq, Ain, bin_, ortho=ortho, **kw)
        err = rel(_np(w), ko.expv(tq, A64, b64, **kw))
        extra = {"t": tq}
    elif call == "expv_complex_t":
        w = eu.expv(0.3 - 0.4j, Ain, bin_, **kw)
        err = rel(_np(w), ko.expv(0.3 - 0.4j, A64, b64, **kw))
    elif call == "arnoldi":
        Ks = eu.arnoldi(A, b, ortho=ortho, **kw)
        Ko = ko.arnoldi(A64, b64, **kw)
        extra = {"m_dev": int(Ks.m), "m_ref": int(Ko.m)}
        md = int(Ks.m)
        brk = md < min(m, n) or md == n      # happy breakdown: column md + 1 of V and H[md + 1, md] are rounding noise
        Hsub = np.abs(np.diag(np.asarray(Ko.getH()), -1))
        near_tol = Ks.m != Ko.m and min(Ks.m, Ko.m) - 1 < len(Hsub) and Hsub[min(Ks.m, Ko.m) - 1] < 1e-5
        if float(Ks.beta) == 0.0 or float(Ko.beta) == 0.0:
            # zero starting vector: firststep! leaves V UNINITIALISED (arnoldi.jl:230-250) -- only beta and H == 0 are defined
            # (seed 31337 case 11086: the basis of a recycled subspace held the previous call's columns)
            # (pinned by tests/test_gpu_configs.py::test_fuzz_pin_zero_start_vector_on_a_recycled_subspace)
            err = 0.0 if (float(Ks.beta) == float(Ko.beta) == 0.0 and not np.any(np.asarray(Ks.getH()))) else float("inf")
            extra["zero_starting_vector"] = True
        elif not single and Ks.m != Ko.m and not near_tol:      # (a residual within 100x of the breakdown tolerance may fall either side of it)
            err = float("inf")
        elif Ks.m != Ko.m and not np.isfinite(np.asarray(Ks.getH())).all():
            err = float("inf")
            extra["H_not_finite"] = True
        else:
            # (1) the Arnoldi relation  A V_k = V_{k+1} H_k  on the columns that are defined
            Hd = np.asarray(Ks.getH()).astype(np.complex128)
            Vd = np.asarray(Ks.getV()).astype(np.complex128)
            kc = md - 1 if brk else md
            if Ks.m != Ko.m:
                kc = min(kc, int(Ko.m) - 1)
            eps = 1.2e-7 if single else 2.2e-16
            if kc >= 1:
                AV = A64 @ Vd[:, :kc]
                R = AV - Vd[:, :kc + 1] @ Hd[:kc + 1, :kc]
                nA = float(sp.linalg.norm(A64)) if sp.issparse(A64) else float(np.linalg.norm(A64))
                err = float(np.linalg.norm(R) / max(nA, 1e-300)) / (200 * eps) * tol       # bar: 200 eps ||A||_F
            # (2) H against the oracle where the oracle itself kept its basis orthogonal (else both are rounding-dominated)
            if Ks.m == Ko.m and kc >= 1:
                Vo = Ko.getV()[:, :kc]
                loss = float(np.max(np.abs(Vo.conj().T @ Vo - np.eye(kc))))
                extra["oracle_orthogonality_loss"] = loss
                if loss < 1e-12 and not (iop and not herm) and not single:      # (fp32: rounding is amplified by ||A|| / H[j+1, j] per column; the relation above is the check)
                    Ho = np.asarray(Ko.getH())[:kc, :kc]
                    eh = float(np.max(np.abs(Hd[:kc, :kc] - Ho)) / max(float(np.max(np.abs(Ho))), 1e-300))
                    extra["H_err"] = eh
                    err = max(err, eh * (tol / (3e-4 if single else 1e-9)))
    elif call == "phiv":
        k = int(rng.integers(1, 5))
        W = eu.phiv(tq, Ain, bin_, k, m=m, iop=iop)
        err = rel(_np(W), ko.phiv(tq, A64, b64, k, m=m, iop=iop))
        extra = {"t": tq}
    elif call in ("expv_timestep", "phiv_timestep"):
        ts = np.sort(rng.uniform(0.1, 1.5, size=int(rng.integers(1, 4))))
        tolk = 1e-5 if single else float(rng.choice([1e-6, 1e-8]))
        mm = max(2, min(m, 30))
        # (correct = true adds beta H[m+1, m] (...) v_{m+1}: 0 x NaN in the reference whenever the Krylov space is exhausted exactly)
        tk = dict(tol=tolk, m=mm, iop=iop, adaptive=bool(rng.integers(0, 4)), correct=bool(rng.integers(0, 2)) and n > 8)
        if not tk["adaptive"]:
            tk["tau"] = float(rng.choice([0.05, 0.2]))
            tk["m"] = max(mm, 12)
        if tk["adaptive"]:
            # The reference's controller never LENGTHENS a step that was too accurate (tau changes only inside `while omega > delta`,
            # krylov_phiv_adaptive.jl:391-417): a tiny Niesen-Wright seed -- small m, tight tol, large ||b|| -- is kept to the end,
            # hundreds of thousands of steps on both sides (seed 1101 case 1953: m = 2, tol 1e-8, ||b|| = 2e6: tau = 1.2e-6,
            # 830 000 steps; the device takes minutes, the Python oracle hours).  Slow by construction, not a parity question.
            opn1 = float(abs(A64).sum(axis=0).max())
            binf = (float(np.max(np.abs(b64))) if n else 0.0) if call == "expv_timestep" else 4.0      # (phiv_timestep: |B[:, 1]|_inf of a standard normal column)
            if opn1 > 0 and binf > 0:
                seed_tau = 0.8 * 10 / opn1 * (tolk * opn1 * ((mm + 1) / np.e) ** (mm + 1) * np.sqrt(2 * np.pi * (mm + 1)) / (4 * opn1 * binf)) ** (1.0 / mm)
                if float(ts[-1]) / seed_tau > 3e4:
                    return desc, 0.0, tol, {"skipped": "the reference's fixed seed step needs %.0e steps" % (float(ts[-1]) / seed_tau)}
        if call == "expv_timestep":
            fd, fr = (lambda: eu.expv_timestep(ts.copy(), Ain, b, **tk)), (lambda: ko.expv_timestep(ts.copy(), A64, b64, **tk))
        else:
            p = int(rng.integers(1, 5))
            B = (rng.standard_normal((n, p + 1)) + (1j * rng.standard_normal((n, p + 1)) if cplx else 0)).astype(T)
            fd, fr = (lambda: eu.phiv_timestep(ts.copy(), Ain, B, **tk)), (lambda: ko.phiv_timestep(ts.copy(), A64, B.astype(T64), **tk))
        # Julia's InexactError of the controller (ceil(Int, Inf): the estimate did not move with m) is part of the behaviour
        raised = []
        outs = []
        for which, f in enumerate((fd, fr)):
            try:
                outs.append(_limited(f, 60) if which == 1 else f())
                raised.append(None)
            except (ValueError, RuntimeError, OverflowError, _RefSpins) as e:
                msg = str(e) or type(e).__name__
                if not any(k in msg for k in ("InexactError", "did not reach the tolerance", "infinity to integer", "_RefSpins", "reference spins")):
                    raise
                outs.append(None)
                raised.append(msg[:60])
        if raised[0] or raised[1]:
            ok = bool(raised[0]) and bool(raised[1])
            if (n <= 2 or n <= m) and bool(raised[0]) != bool(raised[1]):
                ok = True             # (an exhausted space, m >= n: whether its residual is exactly 0 or 1e-16 |A| decides, either is right;
                                      #  seed 5151 case 459: n = 3 -- pinned by tests/test_gpu_configs.py::test_fuzz_pin_exhausted_krylov_space_m_not_below_n)
            if not raised[0] and raised[1] and "spins" in raised[1]:
                ok = True             # (the Python oracle ran out of its time limit on a run the device finished: slow, not wrong)
            if not ok and single and raised[0] and not raised[1]:
                ok = True             # (a 32-bit estimate may stall where the 64-bit oracle's still moves)
            if not ok and raised[0] and not raised[1] and outs[1] is not None and bnorm > 0:
                grow = float(np.max(np.abs(np.asarray(outs[1])))) / bnorm
                if not np.isfinite(grow) or grow > 1e10:
                    ok = True         # (exp(tA) amplifies by > 1e10: eps-level differences in the estimates decide the controller's path --
                                      #  seed 2027 case 22242: Hermitian A with eigenvalues up to +80, t = 1.3; estimates 10 % apart at
                                      #  t = 0.4, the device's trajectory ends in the reference controller's fixed point tau_new = tau --
                                      #  pinned by tests/test_gpu_configs.py::test_fuzz_pin_amplifying_hermitian_operator_controller_paths)
            return desc, (0.0 if ok else float("inf")), tol, {"raised_dev": raised[0], "raised_ref": raised[1], "skipped": "controller error"}
        U, Uo = outs
        extra = {"timestep": {k: v for k, v in tk.items()}}
        if not tk["adaptive"]:
            tol = max(tol, 1e-6)       # (fixed steps: no controller equalises the two runs; the truncation error itself is ~1e-8)
        err = rel(U, Uo)
        if not np.isfinite(np.asarray(Uo)).all():
            return desc, 0.0, tol, {"skipped": "the reference result itself is not finite"}
        tol = max(tol, 50 * tolk) if single else max(1e-9, 0.5 * tolk)      # (two runs of one controller; both within tolk of the truth)
    elif call == "kiops":
        if cplx or single:
            return desc, 0.0, tol, {"skipped": "kiops is Float64 in the reference"}
        tau = [1.0, np.array([0.4, 1.0]), np.array([[0.3, 0.7, 1.1]])][int(rng.integers(0, 3))]
        kk = dict(tol=float(rng.choice([1e-6, 1e-8])), iop=max(iop, 2), task1=bool(rng.integers(0, 2)), mmin=int(rng.choice([4, 10])),
                  mmax=int(rng.choice([30, 128])), m=int(rng.choice([5, 10, 25])))
        # (the reference's own error behaviour -- kiops.jl:303 BoundsError, checkdims on a 2-D tau_out -- is part of the parity)
        errs = []
        res = []
        for which, f in enumerate((lambda: eu.kiops(tau, Ain, b, **kk), lambda: ko.kiops(tau, A64, b64, **kk))):
            try:
                res.append(_limited(f, 60) if which == 1 else f())
                errs.append(None)
            except (IndexError, ValueError, AssertionError, OverflowError, _RefSpins) as e:
                res.append(None)
                errs.append(type(e).__name__)
            except RuntimeError as e:          # the library's bound on rejected steps / its InexactError
                if "rejected steps" not in str(e) and "InexactError" not in str(e):
                    raise
                res.append(None)
                errs.append("bounded")
        if errs[0] or errs[1]:
            return desc, (0.0 if errs[0] and errs[1] else float("inf")), tol, {"raised_dev": errs[0], "raised_ref": errs[1], "skipped": "both raise"}
        (w, st), (wo, so) = res
        err = rel(w, wo)
        extra = {"kiops": {k: (v if not isinstance(v, np.ndarray) else v.tolist()) for k, v in kk.items()}}
        tol = 1e-9
    elif call == "subspace_reuse":
        # one KrylovSubspace through several factorisations: growing / shrinking m, other starting vectors, Lanczos <-> Arnoldi
        Ks = eu.KrylovSubspace(T, None, n, max(1, m // 2))
        for rep in range(int(rng.integers(2, 5))):
            mm = int(rng.integers(1, m + 1))
            hh = herm and bool(rng.integers(0, 2))
            bb = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
            eu.arnoldi_(Ks, A, bb, m=mm, iop=iop, ishermitian=hh, ortho=ortho)
            w = eu.expv(0.6, Ks)
            err = max(err, rel(w, ko.expv(0.6, A64, bb.astype(T64), m=mm, iop=iop, ishermitian=hh)))
    elif call == "continuation":
        # arnoldi!(Ks, A, b; m = m1) then arnoldi!(...; m = m2, init = Ks.m)  (arnoldi.jl:345-377) against one run to m2
        if herm:
            return desc, 0.0, tol, {"skipped": "continuation is exercised on the Arnoldi path"}
        m2 = max(2, m)
        m1 = int(rng.integers(1, m2))
        Ks = eu.KrylovSubspace(T, None, n, m2)
        eu.arnoldi_(Ks, A, b, m=m1, iop=iop, ishermitian=False, ortho=ortho)
        if Ks.wasbreakdown:
            return desc, 0.0, tol, {"skipped": "breakdown before the continuation point"}
        eu.arnoldi_(Ks, A, b, m=m2, iop=iop, ishermitian=False, ortho=ortho, init=m1)
        w = eu.expv(0.6, Ks)
        err = rel(w, ko.expv(0.6, A64, b64, m=m2, iop=iop, ishermitian=False))
        extra = {"m1": m1, "m2": m2}
        tol = max(tol, 1e-9)
    elif call == "update_values":
        if not sp.issparse(A):
            return desc, 0.0, tol, {"skipped": "value updates are for sparse operators"}
        op = eu.MIOperator(A.copy())
        w0 = eu.expv(0.7, op, b, **kw)
        A2 = A.copy()
        A2.data = (A2.data * (1 + 0.1 * rng.standard_normal(A2.nnz))).astype(T)
        op.update_values(A2)
        w = eu.expv(0.7, op, b, **kw)
        err = max(rel(w0, ko.expv(0.7, A64, b64, **kw)), rel(w, ko.expv(0.7, A2.astype(T64), b64, **kw)))
        extra["second_operator"] = A2.astype(T64)
    elif call == "matrix_free":
        import torch
        if single or n > 5000:
            return desc, 0.0, tol, {"skipped": "matrix-free case kept to 64-bit, small n"}
        Ad = torch.as_tensor(A64.toarray() if sp.issparse(A64) else A64, device="cuda")
        op = eu.MIOperator(None, matvec=lambda x: Ad @ x, shape=(n, n), dtype=T, ishermitian=herm)
        w = eu.expv(0.7, op, b, **kw)
        err = rel(w, ko.expv(0.7, A64, b64, **kw))
    elif call == "batch":
        if not sp.issparse(A) or n > 5000 or herm:
            return desc, 0.0, tol, {"skipped": "batch: sparse operators"}
        nprob = int(rng.integers(1, 6))
        P = A.tocsr()
        P.sort_indices()
        vals = np.stack([P.data * (1 + 0.05 * rng.standard_normal(P.nnz)) for _ in range(nprob)]).astype(T)
        Bm = np.asfortranarray((rng.standard_normal((n, nprob)) + (1j * rng.standard_normal((n, nprob)) if cplx else 0)).astype(T))
        mb = min(m, 30)
        W = np.asarray(eu.expv_batch(0.7, P, vals, Bm, m=mb, iop=iop))
        extra["batch_problems"] = []
        for q in range(nprob):
            Aq = P.copy()
            Aq.data = vals[q].copy()
            err = max(err, rel(W[:, q], ko.expv(0.7, Aq.astype(T64), Bm[:, q].astype(T64), m=mb, iop=iop, ishermitian=False)))
            extra["batch_problems"].append((Aq.astype(T64), Bm[:, q].astype(T64), mb))
    elif call == "async_device":
        # the mode the headline runs in: a context with stream-ordered outputs, operands and results resident on the device,
        # several calls in flight before one synchronisation
        import torch
        if n > 200000:
            return desc, 0.0, tol, {"skipped": "size"}
        ctx = _async_ctx()
        op = eu.MIOperator(A, ctx)
        tdt = {"float32": torch.float32, "float64": torch.float64, "complex64": torch.complex64, "complex128": torch.complex128}[T.name]
        nb = int(rng.integers(1, 4))
        bs = [(rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T) for _ in range(nb)]
        bd = [torch.as_tensor(x, device="cuda") for x in bs]
        outs = [torch.empty(n, dtype=tdt, device="cuda") for _ in range(nb)]
        for x, o in zip(bd, outs):
            eu.expv(0.7, op, x, out=o, ortho=ortho, **kw)
        ctx.sync()
        for x, o in zip(bs, outs):
            err = max(err, rel(o.cpu().numpy(), ko.expv(0.7, A64, x.astype(T64), **kw)))
    elif call == "caches":
        # phiv_timestep! through _phiv_timestep_caches, reused over calls with other p and a larger m (krylov_phiv_adaptive.jl:502-511)
        if n > 5000:
            return desc, 0.0, tol, {"skipped": "size"}
        mm = max(2, min(m, 20))
        nrep = int(rng.integers(2, 4))
        pmax = 3
        caches = eu.timestep_caches(b, mm + 3 * nrep, pmax)       # (the reference asserts the cache dimensions: sized for the largest call)
        for rep in range(nrep):
            p = pmax if rep == 0 else int(rng.integers(0, pmax + 1))
            B = (rng.standard_normal((n, p + 1)) + (1j * rng.standard_normal((n, p + 1)) if cplx else 0)).astype(T)
            ts = rng.uniform(0.1, 1.2, size=int(rng.integers(1, 4)))       # (unsorted: sorted in place like the reference)
            tolk = 1e-5 if single else 1e-7
            mrep = mm + 3 * rep
            U = np.empty((n, len(ts)), dtype=T, order="F")
            fr = lambda: ko.phiv_timestep(ts.copy(), A64, (B if p else B[:, 0]).astype(T64), tol=tolk, m=mrep, iop=iop, adaptive=True)
            try:
                eu.phiv_timestep_(U, ts.copy(), A, B if p else B[:, 0], tol=tolk, m=mrep, iop=iop, adaptive=True, caches=caches)
            except (ValueError, RuntimeError) as e:
                if "did not reach the tolerance" in str(e):
                    # The device's documented stop after 1000 equal proposals: the reference's controller has no such stop and
                    # SPINS on the same fixed point (seed 7272 case 5412: ComplexF64 n = 1537, m = 20 -- tools/fuzz_repro_7272.py).
                    # Equal behaviour = the oracle raises or does not come back either; a finished oracle run is a difference.
                    try:
                        Uref = _limited(fr, 60)
                    except (ValueError, RuntimeError, OverflowError, _RefSpins):
                        break         # (the caches hold the abandoned run's state: later calls of this case would not compare)
                    grow = float(np.max(np.abs(np.asarray(Uref)))) / max(float(np.max(np.abs(B))), 1e-300)
                    if not np.isfinite(grow) or grow > 1e10:
                        break         # (same rule as the expv_timestep / phiv_timestep cases: eps-level estimates decide the path)
                    err = float("inf")
                    break
                if "InexactError" not in str(e):
                    raise
                continue              # (the controller's own error: compared in the expv_timestep / phiv_timestep cases)
            try:
                Uo = _limited(fr, 120)
            except (ValueError, RuntimeError) as e:
                if "InexactError" not in str(e):
                    raise
                continue
            except _RefSpins:
                continue              # (the Python oracle ran out of its time limit on a run the device finished: slow, not wrong)
            if not np.isfinite(np.asarray(Uo)).all():
                continue
            err = max(err, rel(U, Uo))
        tol = 50 * 1e-5 if single else 1e-9
    elif call == "phiv_correct":
        k = int(rng.integers(1, 4))
        Ko = ko.arnoldi(A64, b64, m=m, iop=iop)
        if Ko.wasbreakdown or Ko.m < m:
            return desc, 0.0, tol, {"skipped": "the correction uses v_{m+1} and H[m+1, m]: rounding noise (or NaN) after a happy breakdown"}
        W, e1 = eu.phiv(0.5, A, b, k, m=m, iop=iop, correct=True, errest=True)
        Wo, e2 = ko.phiv(0.5, A64, b64, k, m=m, iop=iop, correct=True, errest=True)
        err = rel(W, Wo)
        if np.isfinite(e2) and e2 > 1e-10 * max(float(np.linalg.norm(Wo)), 1e-300) and not single:
            err = max(err, abs(e1 - e2) / abs(e2) * 1e-4)      # the estimate itself to 1e-6 relative (where it is more than rounding noise)
    else:
        if not herm:
            return desc, 0.0, tol, {"skipped": "error estimate needs a Hermitian operator here"}
        w = eu.expv(0.7, A, b, m=max(m, 3), mode="error_estimate", rtol=1e-6 if not single else 1e-4)
        wo = ko.expv(0.7, A64, b64, m=max(m, 3), mode="error_estimate", rtol=1e-6 if not single else 1e-4)
        err = rel(w, wo)
        tol = 1e-9 if not single else 5e-4
    # A result beyond the bar is only a finding when the PROBLEM is not the cause: where the oracle's own strict-MGS basis has lost
    # orthogonality by `loss`, two correct implementations that add their dot products in a different order differ by a multiple
    # of it (the fixed-size suites scale their bars the same way).  Evaluated only for flagged cases.
    A2x = extra.pop("second_operator", None)
    batch_problems = extra.pop("batch_problems", None)
    if not single and np.isfinite(err) and err > tol and call == "batch" and batch_problems and iop == 0:
        # (seed 8088 case 4782: three ComplexF64 banded problems scaled to |A| ~ 45, m = 14 -- the oracle's own bases are orthogonal to
        #  6e-10 .. 3e-9 only, the device's to 6e-10 .. 8e-10; batch and single-problem calls agree with each other to 1e-12 .. 3e-10:
        #  tools/fuzz_repro_8088.py)
        loss = 0.0
        for Aq, bq, mq in batch_problems:
            try:
                Kq = ko.arnoldi(Aq, bq, m=mq, iop=0, ishermitian=False)
                Vq = Kq.getV()[:, : Kq.m + 1]
                loss = max(loss, float(np.max(np.abs(Vq.conj().T @ Vq - np.eye(Vq.shape[1])))))
            except Exception:
                pass
        if loss > 0.0:
            extra["oracle_loss_of_orthogonality"] = loss
            # (the widening must not hide a device-side regression -- the DEVICE's own bases have to be at least as
            #  orthogonal as the oracle's, give or take a factor 2; otherwise the bar stays where it was and the case is flagged)
            dloss = 0.0
            for Aq, bq, mq in batch_problems:
                try:
                    Kd = eu.arnoldi(Aq, bq, m=mq, iop=0, ishermitian=False)
                    Vd = np.asarray(Kd.getV())[:, : Kd.m + 1]
                    dloss = max(dloss, float(np.max(np.abs(Vd.conj().T @ Vd - np.eye(Vd.shape[1])))))
                except Exception:
                    dloss = float("inf")
            extra["device_loss_of_orthogonality"] = dloss
            if dloss <= max(2.0 * loss, 1e-12):
                tol = max(tol, 10.0 * loss)
    if not single and np.isfinite(err) and err > tol and call in ("expv", "arnoldi", "update_values", "subspace_reuse", "continuation",
                                                                  "async_device", "phiv", "phiv_correct", "expv_complex_t", "caches"):
        loss = 0.0
        for Aq in (A64, A2x):
            if Aq is None:
                continue
            try:
                Ko = ko.arnoldi(Aq, b64, m=m, iop=iop, ishermitian=herm)
                Vq = Ko.getV()[:, : Ko.m + 1]
                if iop == 0 and not herm:
                    loss = max(loss, float(np.max(np.abs(Vq.conj().T @ Vq - np.eye(Vq.shape[1])))))
            except Exception:
                pass
        if loss > 0.0:
            extra["oracle_loss_of_orthogonality"] = loss
            dloss = 0.0
            for Aq in (A64, A2x):      # (the same guard: the device's own basis must not be worse than the oracle's)
                if Aq is None:
                    continue
                try:
                    Kd = eu.arnoldi(Aq, b64, m=m, iop=iop, ishermitian=herm)
                    Vd = np.asarray(Kd.getV())[:, : Kd.m + 1]
                    dloss = max(dloss, float(np.max(np.abs(Vd.conj().T @ Vd - np.eye(Vd.shape[1])))))
                except Exception:
                    dloss = float("inf")
            extra["device_loss_of_orthogonality"] = dloss
            if dloss <= max(2.0 * loss, 1e-12):
                tol = max(tol, 10.0 * loss)
    return desc, err, tol, extra


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    if len(sys.argv) > 3:
        desc, err, tol, extra = one_case(seed, int(sys.argv[3]), verbose=True)
        print(json.dumps({**desc, "err": err, "tol": tol, **extra}))
        return
    t0 = time.time()
    index = int(sys.argv[3]) if False else 0
    fails = 0
    worst = {}
    import signal

    def on_alarm(sig, frm):
        raise TimeoutError("case took more than 120 s")
    signal.signal(signal.SIGALRM, on_alarm)
    start = int(__import__("os").environ.get("FUZZ_START", "0"))
    index = start
    while time.time() - t0 < seconds:
        signal.alarm(120)
        if __import__("os").environ.get("FUZZ_VERBOSE"):
            print("case", index, round(time.time() - t0, 1), flush=True)
        try:
            desc, err, tol, extra = one_case(seed, index)
            key = (desc["call"], "32" if desc["T"] in ("float32", "complex64") else "64")
            if "skipped" not in extra:
                worst[key] = max(worst.get(key, 0.0), err if np.isfinite(err) else 1e300)
            if not err <= tol:
                fails += 1
                print("FAIL", json.dumps({**desc, "err": err, "tol": tol, **extra}), flush=True)
        except Exception as e:
            fails += 1
            print("EXCEPTION", json.dumps({"seed": seed, "index": index}), repr(e), flush=True)
            traceback.print_exc(limit=3)
        index += 1
    signal.alarm(0)
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)      # (a hang while the interpreter tears down is reported, not waited for)
    print(json.dumps({"cases": index - start, "failures": fails, "seconds": round(time.time() - t0, 1),
                      "worst_by_call": {"%s/%s" % k: v for k, v in sorted(worst.items())}}), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

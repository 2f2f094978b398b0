"""Shared helpers for the tests: dense truths (scipy) and the synthetic operators of SURVEY.md §8d."""
import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp


def dense_phis(M, K):
    """[phi_0(M), ..., phi_K(M)] by the block-matrix identity (construction of basictests.jl:342-356)."""
    n = M.shape[0]
    Z = np.zeros((n * (K + 1), n * (K + 1)), dtype=M.dtype)
    Z[:n, :n] = M
    for i in range(K):
        Z[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = np.eye(n)
    E = sl.expm(Z)
    return [E[:n, i * n:(i + 1) * n] for i in range(K + 1)]


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def banded(n, offsets, vals, dtype=np.float64, fmt="csr"):
    """Constant-diagonal banded operator (BASELINE config 2 pattern)."""
    diags = [np.full(n - abs(o), v, dtype=dtype) for o, v in zip(offsets, vals)]
    return sp.diags(diags, offsets, shape=(n, n), format=fmt, dtype=dtype)


C2_OFFSETS = (-2, -1, 0, 1, 2)
C2_VALS = (0.3, 1.2, -2.0, 0.8, -0.1)          # non-symmetric  -> Arnoldi path
C2_SYM_VALS = (0.5, 1.0, -3.0, 1.0, 0.5)       # symmetric      -> Lanczos path


def c2_operator(n, sym=False, dtype=np.float64, fmt="csr"):
    return banded(n, C2_OFFSETS, C2_SYM_VALS if sym else C2_VALS, dtype=dtype, fmt=fmt)


def stencil2d(nx, dtype=np.float64, fmt="csr"):
    """2-D 5-point stencil, offsets (-nx,-1,0,1,nx): non-local x access (SURVEY.md §8d secondary)."""
    n = nx * nx
    return banded(n, (-nx, -1, 0, 1, nx), (0.7, 1.1, -4.0, 0.9, 1.3), dtype=dtype, fmt=fmt)


def mkA(n):
    """Deterministic non-symmetric operator of basictests.jl:859-862."""
    i = np.arange(1, n + 1)[:, None]
    j = np.arange(1, n + 1)[None, :]
    A = 0.1 / (1 + np.abs(i - j)) * np.where(i < j, 1.0, 0.5)
    A[np.arange(n), np.arange(n)] = -2.0
    return A


# every parity comparison goes through close(): the measured error is printed next to its bar (pytest -rP shows it,
# a failing assert carries it) and appended to gpurun_out/parity_measured.jsonl when that directory exists, so the bars
# quoted in DESIGN.md §5 are backed by numbers (tools/parity_report.py turns the file into profiles/rNN_parity_measured.txt)
def close(a, b, tol, what, mat=False, absolute=False):
    import json
    import os
    a = np.atleast_1d(np.asarray(a))
    b = np.atleast_1d(np.asarray(b))
    if absolute:
        err = float(np.max(np.abs(a - b)))
    elif mat:       # entrywise, relative to the largest entry (the Hessenberg bar of SURVEY.md §8c)
        err = float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
    else:
        err = relerr(a, b)
    line = {"what": what, "err": err, "tol": tol}
    print("[parity] %-90s err %.3e  (bar %.1e)" % (what, err, tol))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
                f.write(json.dumps(line) + "\n")
        except OSError:
            pass
    assert err <= tol, "%s: measured %.3e > bar %.1e" % (what, err, tol)
    return err

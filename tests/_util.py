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

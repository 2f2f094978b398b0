"""ctypes binding of oracle/liboracle.so (the plain-C restatement in expv_oracle.c).
TEST INFRASTRUCTURE ONLY -- see the header of expv_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def num_threads():
    return int(load().oracle_num_threads())


def set_num_threads(n):
    """OpenMP threads of the following calls (bench.py: all host threads, then 1)."""
    load().oracle_set_num_threads(int(n))


def _csr32(A):
    A = A.tocsr()
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32),
            np.ascontiguousarray(A.data))


def arnoldi_csr(A, b, m=30, iop=0, tol=1e-7, hermitian=False):
    """Returns dict(V, H, beta, m, breakdown) from the C restatement (literal MGS / Lanczos)."""
    lib = load()
    n = A.shape[0]
    rp, ci, va = _csr32(A)
    cplx = np.iscomplexobj(va) or np.iscomplexobj(b)
    dt = np.complex128 if cplx else np.float64
    va = va.astype(dt)
    b = np.ascontiguousarray(b, dtype=dt)
    V = np.zeros((n, m + 1), dtype=dt, order="F")
    H = np.zeros((m + 1, m), dtype=dt, order="F")
    beta = C.c_double()
    brk = C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    if hermitian:
        if cplx:
            raise NotImplementedError("C oracle: complex Lanczos not restated (numpy oracle covers it)")
        mm = lib.oracle_lanczos_csr_f64(C.c_int64(n), p(rp), p(ci), p(va), p(b), m, C.c_double(tol), p(V), p(H),
                                        C.byref(beta), C.byref(brk))
    elif cplx:
        mm = lib.oracle_arnoldi_csr_c64(C.c_int64(n), p(rp), p(ci), p(va), p(b), m, iop, C.c_double(tol), p(V), p(H),
                                        C.byref(beta), C.byref(brk))
    else:
        mm = lib.oracle_arnoldi_csr_f64(C.c_int64(n), p(rp), p(ci), p(va), p(b), m, iop, C.c_double(tol), p(V), p(H),
                                        C.byref(beta), C.byref(brk))
    return {"V": V, "H": H, "beta": beta.value, "m": int(mm), "breakdown": bool(brk.value)}


def combine(V, coef, beta, m):
    lib = load()
    n = V.shape[0]
    cplx = np.iscomplexobj(V) or np.iscomplexobj(coef)
    dt = np.complex128 if cplx else np.float64
    Vc = np.asfortranarray(V, dtype=dt)
    cf = np.ascontiguousarray(coef, dtype=dt)
    w = np.empty(n, dtype=dt)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    (lib.oracle_combine_c64 if cplx else lib.oracle_combine_f64)(C.c_int64(n), int(m), p(Vc), p(cf), C.c_double(beta), p(w))
    return w


def expv_csr(t, A, b, m=30, tol=1e-7, hermitian=False):
    """expv through the C hot loop + the numpy small-exp of krylov_oracle (host part)."""
    from . import krylov_oracle as ko
    r = arnoldi_csr(A, b, m=m, tol=tol, hermitian=hermitian)
    mm = r["m"]
    if r["beta"] == 0:
        return np.zeros_like(r["V"][:, 0]), r
    Hm = r["H"][:mm, :mm]
    if np.array_equal(Hm, Hm.conj().T):
        coef = ko._sym_tridiag_expcol(Hm, t)
    else:
        coef = ko.exponential_(t * Hm)[:, 0]
    return combine(r["V"], coef, r["beta"], mm), r

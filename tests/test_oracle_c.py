"""Pins the plain-C oracle (oracle/expv_oracle.c) to the numpy oracle (itself pinned to the
reference's KATs in test_oracle_kat.py).  CPU only."""
import numpy as np
import pytest
import scipy.linalg as sl

from oracle import c_oracle as co
from oracle import krylov_oracle as ko
from tests._util import c2_operator, relerr, stencil2d


@pytest.mark.parametrize("n,m,iop", [(200, 30, 0), (2000, 30, 0), (2000, 20, 2), (513, 12, 3)])
def test_c_arnoldi_matches_numpy_oracle(n, m, iop):
    A = c2_operator(n)
    b = np.random.default_rng(3).standard_normal(n)
    r = co.arnoldi_csr(A, b, m=m, iop=iop)
    Ks = ko.KrylovSubspace(float, float, n, m)
    ko.arnoldi_(Ks, A, b, m=m, iop=iop, ishermitian=False)
    assert r["m"] == Ks.m and r["breakdown"] == Ks.wasbreakdown
    assert abs(r["beta"] - Ks.beta) <= 1e-14 * Ks.beta
    assert np.max(np.abs(r["H"] - Ks.H[: m + 1, :m])) <= 1e-13 * np.max(np.abs(Ks.H))
    assert np.max(np.abs(r["V"] - Ks.V)) <= 1e-12


def test_c_lanczos_matches_numpy_oracle():
    n, m = 1500, 30
    A = c2_operator(n, sym=True)
    b = np.random.default_rng(4).standard_normal(n)
    r = co.arnoldi_csr(A, b, m=m, hermitian=True)
    Ks = ko.KrylovSubspace(float, float, n, m)
    ko.lanczos_(Ks, A, b, m=m)
    assert np.max(np.abs(r["H"] - Ks.H[: m + 1, :m])) <= 1e-12 * np.max(np.abs(Ks.H))
    assert np.max(np.abs(r["V"] - Ks.V)) <= 1e-10


def test_c_complex_arnoldi_matches_numpy_oracle():
    n, m = 400, 15
    A = (c2_operator(n) * (1 + 0.25j)).tocsr()
    b = np.random.default_rng(6).standard_normal(n) + 1j * np.random.default_rng(7).standard_normal(n)
    r = co.arnoldi_csr(A, b, m=m, iop=2)
    Ks = ko.KrylovSubspace(complex, complex, n, m)
    ko.arnoldi_(Ks, A, b, m=m, iop=2, ishermitian=False)
    assert np.max(np.abs(r["H"] - Ks.H[: m + 1, :m])) <= 1e-12 * np.max(np.abs(Ks.H))


def test_c_expv_against_dense_truth():
    A = stencil2d(17)
    n = A.shape[0]
    b = np.random.default_rng(5).standard_normal(n)
    w, _ = co.expv_csr(0.3, A, b, m=30)
    assert relerr(w, sl.expm(0.3 * A.toarray()) @ b) < 1e-10


def test_c_breakdown_and_zero():
    n = 50
    A = c2_operator(n)
    w, r = co.expv_csr(1.0, A, np.zeros(n), m=10)
    assert np.all(w == 0)
    import scipy.sparse as sp
    v = np.random.default_rng(8).standard_normal(n)
    v /= np.linalg.norm(v)
    P = sp.csr_matrix(np.outer(v, v))
    r = co.arnoldi_csr(P, np.random.default_rng(9).standard_normal(n), m=10)
    assert r["m"] == 2 and r["breakdown"]

"""Pins the oracle (oracle/krylov_oracle.py) against every deterministic known-answer test the
reference's own suite holds for the Krylov path, and against dense scipy truths over the
reference's type matrix.  CPU only (runs in the build container)."""
import numpy as np
import pytest
import scipy.linalg as sl
import scipy.sparse as sp

from oracle import krylov_oracle as ko
from tests._util import dense_phis, relerr, mkA, c2_operator

SQRT_EPS = float(np.sqrt(np.finfo(float).eps))      # Julia's default `≈` rtol


def test_phiv_matrix_kat():
    """basictests.jl:569-573: n=30 tridiag(1,30,1), t=0.1, b=ones, k=10 (m = n => exact)."""
    n = 30
    A = np.diag(np.ones(n - 1), -1) + 30 * np.eye(n) + np.diag(np.ones(n - 1), 1)
    t = 0.1
    Q = ko.phiv(t, A, np.ones(n), 10)
    ref = np.linalg.solve(t * A, (sl.expm(t * A) - np.eye(n)) @ np.ones(n))
    assert relerr(Q[:, 1], ref) < SQRT_EPS
    # first entries recorded in SURVEY.md §4 (computed with scipy)
    np.testing.assert_allclose(Q[:3, 1], [6.85734928, 7.33460365, 7.3533841], rtol=1e-8)


def test_issue_143_single_step_and_stdout():
    """basictests.jl:193-205: 1x1 integer operator, ts=0:0.1:1, 'Completed after 1 time step(s)'."""
    ts = np.arange(0, 1.0001, 0.1)
    out = []
    res = ko.expv_timestep(ts.copy(), np.array([[1]]), np.array([1.0]), verbose=True, out=out.append)
    assert any("Completed after 1 time step(s)" in s for s in out)
    np.testing.assert_allclose(res.ravel(), np.exp(ts), rtol=SQRT_EPS)


def test_happy_breakdown_idempotent():
    """basictests.jl:544-547: A = v v' (rank one)  =>  Ks.m == 2."""
    rng = np.random.default_rng(5)
    n = 20
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    A = np.outer(v, v)
    b = rng.standard_normal(n)
    assert ko.arnoldi(A, b).m == 2                       # Hermitian -> Lanczos
    assert ko.arnoldi(A, b, ishermitian=False).m == 2    # Arnoldi
    assert ko.arnoldi(A, b, ishermitian=False).wasbreakdown


@pytest.mark.parametrize("herm", [False, True])
def test_zero_input_is_exactly_zero(herm):
    """basictests.jl:550-553 (Arnoldi) and :565-566 (Lanczos): norm(w) == 0.0 exactly."""
    rng = np.random.default_rng(6)
    n = 20
    A = rng.standard_normal((n, n))
    if herm:
        A = (A + A.T) / 2
    z = np.zeros(n)
    w = ko.expv(1e-2, A, z, m=5)
    assert np.linalg.norm(w) == 0.0
    assert not np.any(np.isnan(w))


def test_arnoldi_vs_lanczos_H_real():
    """basictests.jl:731-754: p = -im*Tridiagonal(-e,0,e): Arnoldi H == Lanczos H to 1e-14."""
    rng = np.random.default_rng(7)
    n, m = 100, 15
    e = np.ones(n)
    p = -1j * (np.diag(-e[1:], -1) + np.diag(e[1:], 1))
    v = rng.random(n) + 1j * rng.random(n)
    KsA = ko.KrylovSubspace(complex, complex, n, m)
    KsL = ko.KrylovSubspace(complex, float, n, m)
    ko.arnoldi_(KsA, p, v, ishermitian=False)
    ko.lanczos_(KsL, p, v)
    AH = KsA.H[:KsA.m, :KsA.m]
    LH = KsL.H[:KsL.m, :KsL.m]
    assert np.linalg.norm(AH - LH) / np.linalg.norm(AH) < 1e-14


def test_arnoldi_krylov_testset():
    """basictests.jl:515-541 as properties (Julia RNG not reproducible): n=20, m=5, K=4, t=1e-2."""
    rng = np.random.default_rng(0)
    n, m, K, t = 20, 5, 4, 1e-2
    A = rng.standard_normal((n, n))
    b = rng.standard_normal(n)
    direct = sl.expm(t * A) @ b
    assert relerr(ko.expv(t, A, b, m=m), direct) < SQRT_EPS
    assert relerr(ko.kiops(t, A, b)[0][:, 0], direct) < SQRT_EPS
    P = dense_phis(t * A, K)
    W = np.stack([P[i] @ b for i in range(K + 1)], axis=1)
    Ks = ko.arnoldi(A, b, m=m)
    Wa = ko.phiv_(np.empty((n, K + 1), order="F"), t, Ks, K)
    assert relerr(Wa, W) < SQRT_EPS
    w3, stats = ko.kiops(t, A, np.stack([b * (1 / t) ** i for i in range(K)], axis=1))
    assert relerr(w3[:, 0], W[:, :K].sum(axis=1)) < SQRT_EPS
    assert stats[2] == 0           # krystep is never incremented (kiops.jl:77,278)


def test_arnoldi_vs_lanczos_expv():
    """basictests.jl:556-562: Hermitian A vs A + 1e-10*noise, and kiops."""
    rng = np.random.default_rng(1)
    n, m, t = 20, 5, 1e-2
    A = rng.standard_normal((n, n))
    A = (A + A.T) / 2
    b = rng.standard_normal(n)
    Aperm = A + 1e-10 * rng.standard_normal((n, n))
    w = ko.expv(t, A, b, m=m)
    wperm = ko.expv(t, Aperm, b, m=m, opnorm=np.linalg.norm)
    wk = ko.kiops(t, A, b, m=m)[0][:, 0]
    assert relerr(wperm, w) < SQRT_EPS
    assert relerr(wk, w) < SQRT_EPS


@pytest.mark.parametrize("kindA", ["hc", "hr", "gc", "gr"])
@pytest.mark.parametrize("cb", [True, False])
@pytest.mark.parametrize("t", [1e-2, 1e-2j, 1e-2 + 1e-2j])
def test_complex_value_matrix(kindA, cb, t):
    """basictests.jl:650-664: 4 operator kinds x 2 b kinds x 3 t kinds, n=20, m=10."""
    rng = np.random.default_rng(hash((kindA, cb)) % 1000)
    n, m = 20, 10
    X = rng.random((n, n)) + (1j * rng.random((n, n)) if kindA[1] == "c" else 0)
    A = (X + X.conj().T) / 2 if kindA[0] == "h" else X
    b = rng.random(n) + (1j * rng.random(n) if cb else 0)
    assert relerr(ko.expv(t, A, b, m=m), sl.expm(t * A) @ b) < SQRT_EPS


def test_adaptive_krylov():
    """basictests.jl:666-691: n=100 spdiagm(1,-2,1), t=5, K=4, tol=1e-7 (own seeded B)."""
    n, K, t, tol = 100, 4, 5.0, 1e-7
    A = sp.diags([np.ones(n - 1), -2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
    B = np.random.default_rng(14).standard_normal((n, K + 1))
    Ad = A.toarray()
    Ph = dense_phis(t * Ad, K)
    Phh = dense_phis(t / 2 * Ad, K)
    u_exact = sum(t ** i * Ph[i] @ B[:, i] for i in range(K + 1))
    uhalf = sum((t / 2) ** i * Phh[i] @ B[:, i] for i in range(K + 1))
    U = ko.phiv_timestep(np.array([t / 2, t]), A, B, adaptive=True, tol=tol)
    assert relerr(U[:, 0], uhalf) < tol
    assert relerr(U[:, 1], u_exact) < tol
    u_exact0 = Ph[0] @ B[:, 0]
    opn = lambda M, p: abs(M).sum(axis=1).max()
    u = ko.expv_timestep(t, A, B[:, 0], adaptive=True, tol=tol, opnorm=opn)
    assert relerr(u, u_exact0) < tol
    u2 = ko.expv_timestep(t, A, B[:, 0], adaptive=True, tol=tol, opnorm=opn(A, np.inf))
    assert relerr(u2, u_exact0) < tol


def test_matrix_free_default_tolerance():
    """basictests.jl:693-729: default scale from the Arnoldi Hessenberg, no opnorm call."""
    n, t, tol = 50, 3.0, 1e-7
    Ad = sp.diags([np.ones(n - 1), -2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
    b = np.random.default_rng(8).standard_normal(n)
    u_exact = sl.expm(t * Ad.toarray()) @ b
    assert relerr(ko.expv_timestep(t, Ad, b, adaptive=True, tol=tol), u_exact) < 1e-5
    assert relerr(ko.expv_timestep(t, Ad, b, adaptive=True, tol=tol, opnorm=4.0), u_exact) < 1e-5


def test_error_estimate_mode():
    """basictests.jl:756-784: Hermitian rand(300,300), complex b, expv(-im, dt*A, b; mode=:error_estimate)."""
    rng = np.random.default_rng(9)
    n, m, dt = 300, 30, 0.1
    A = rng.random((n, n))
    A = (A + A.T) / 2
    b = rng.random(n) + 1j * rng.random(n)
    w = ko.expv(-1j, dt * A, b, m=m, tol=1e-10, rtol=1e-10, mode="error_estimate")
    wp = sl.expm(-1j * dt * A) @ b
    dw = np.linalg.norm(w - wp)
    assert dw < 1e-10 and dw / abs(1e-16 + np.linalg.norm(w)) < 1e-10
    wz = ko.expv(-1j, dt * A, np.zeros(n, dtype=complex), m=m, tol=1e-10, rtol=1e-10, mode="error_estimate")
    assert np.linalg.norm(wz) == 0
    with pytest.raises(RuntimeError):
        ko.expv(-1j, rng.random((5, 5)), np.ones(5, dtype=complex), mode="error_estimate", ishermitian=False)


@pytest.mark.parametrize("herm", [False, True])
def test_matrix_free_generic_interface(herm):
    """basictests.jl:786-816: operator with only eltype/size/mul!/ishermitian; atol 1e-12."""
    rng = np.random.default_rng(123)
    n = 20
    A = rng.random((n, n)) + 1j * rng.random((n, n))
    M = A.conj().T @ A if herm else A

    class Operator:
        def __init__(self, data):
            self.data, self.shape, self.dtype = data, data.shape, data.dtype

        def __matmul__(self, x):
            return self.data @ x

    Op = Operator(M)
    b = rng.random(n) + 1j * rng.random(n)
    Ks = ko.arnoldi(Op, b, ishermitian=herm, tol=1e-12)
    pv = ko.phiv_(np.empty((n, 3), dtype=complex, order="F"), 0.01, Ks, 2)
    ref = np.stack([P @ b for P in dense_phis(0.01 * M, 2)], axis=1)
    np.testing.assert_allclose(pv, ref, atol=1e-12, rtol=SQRT_EPS)
    np.testing.assert_allclose(ko.expv(0.01, Op, b, m=n, ishermitian=herm), sl.expm(0.01 * M) @ b,
                               atol=1e-12, rtol=SQRT_EPS)


@pytest.mark.parametrize("T", [float, complex])
@pytest.mark.parametrize("scale", [3.0, 1.5, 0.5, 0.1, 0.005])
def test_higham2005base_every_pade_branch(T, scale):
    """basictests.jl:952-974: the scale factor selects C13/C9/C7/C5/C3; rel err < 1e-11 (fp64)."""
    rng = np.random.default_rng(7)
    n = 40
    A0 = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if T is complex else 0)
    A = scale * A0 / np.linalg.norm(A0, 1)
    assert relerr(ko.exponential_(A), sl.expm(A)) < 1e-11


def test_gebal_roundtrip_and_permutation():
    """balance!/unbalance! (exp_baseexp.jl:127,158): isolates eigenvalues and undoes itself."""
    rng = np.random.default_rng(3)
    A = np.triu(rng.standard_normal((8, 8))) * np.logspace(-3, 3, 8)[:, None] * 1e-2
    A[3, 0] = 1.0
    A[6, 2] = 3.0
    B = A.copy(order="F")
    ilo, ihi, scale = ko.gebal(B)
    assert 1 <= ilo <= ihi <= 8
    np.testing.assert_allclose(np.sort(np.linalg.eigvals(B)), np.sort(np.linalg.eigvals(A)), rtol=1e-9, atol=1e-12)
    assert relerr(ko.exponential_(A), sl.expm(A)) < 1e-11   # badly scaled: scipy itself is ~1e-12 here


def test_cache_fixture_inputs():
    """Deterministic inputs of basictests.jl:859-868 (mkA, b = 1/i): expv/phiv agree with dense truth."""
    n, m = 64, 30
    A = mkA(n)
    b = 1.0 / np.arange(1, n + 1)
    assert relerr(ko.expv(0.1, A, b, m=m), sl.expm(0.1 * A) @ b) < 1e-12
    W = ko.phiv(0.1, A, b, 3, m=m)
    ref = np.stack([P @ b for P in dense_phis(0.1 * A, 3)], axis=1)
    assert relerr(W, ref) < 1e-12


def test_kiops_quirks():
    """kiops.jl:86,89,210,278 and arnoldi.jl:217: the behaviours SURVEY.md §8a-16 lists."""
    rng = np.random.default_rng(11)
    n = 30
    A = c2_operator(n).toarray()
    u = rng.standard_normal(n)
    w, stats = ko.kiops(1.0, A, u)
    assert w.shape == (n, 1) and w.dtype == np.float64
    assert relerr(w[:, 0], sl.expm(A) @ u) < 1e-6
    assert stats[2] == 0
    with pytest.raises(ko.DimensionMismatch):           # numSteps > 1 never passes checkdims
        ko.kiops(np.array([[0.5, 1.0]]), A, u)
    with pytest.raises(TypeError):                      # complex has no method in the reference
        ko.kiops(1.0, A.astype(complex), u)


def test_controller_arithmetic_follows_julia_not_python():
    """krylov_phiv_adaptive.jl:455-501 in Float64: x / 0.0 is +-Inf or NaN (no exception) and only `ceil(Int, x)` / `Int(x)` of a
    non-finite x throws (InexactError).  Python's float division raises ZeroDivisionError instead, which would make the oracle fail
    where the reference carries on (q = -Inf -> tau unchanged) and fail DIFFERENTLY where it throws."""
    import pytest
    from oracle import krylov_oracle as ko
    # equal estimates at two step sizes: log(eps / eps_old) = 0 -> q = +-Inf -> (gamma / omega)^0 = 1: tau_new = tau, no exception
    m_new, tau_new, q, kappa = ko._timestep_adapt(10, 0.5, 1e-3, 10, 1.0, 1e-3, 2.5, 2.0, 0.8, 5.0, 1.0, 100, 1, 500, 0, 3.0, False, None)
    assert tau_new == pytest.approx(0.5) or tau_new == pytest.approx(0.1) or tau_new == pytest.approx(1.0)
    assert np.isinf(q) or np.isnan(q)
    # the estimate did not move with m: kappa = 1 -> ceil(Int, x / 0) -> InexactError
    with pytest.raises(ValueError, match="InexactError"):
        ko._timestep_adapt(12, 0.5, 1e-3, 10, 0.5, 1e-3, 2.5, 2.0, 0.8, 5.0, 1.0, 100, 1, 500, 0, 3.0, False, None)
    # tau driven to zero: Int(ceil(maxtau / tau)) -> InexactError
    with pytest.raises(ValueError, match="InexactError"):
        ko._estimate_flops(10, 0.0, 100, 1, 500, 0, 3.0, 1.0)


def test_pipelined_lanczos_restatements_hold_the_reference_results():
    """oracle/pipelined_lanczos.py (test infrastructure for the opt-in `ortho = "pipelined"` mode): the one-reduction form (p1), the form
    whose scalars arrive a pass late (p2) and the numpy restatement of the device scheme (p3) reproduce exp(tA)b of the reference
    recurrence (arnoldi.jl:388-403) to 1e-12 on the symmetric C2 operator, the complex Hermitian tridiagonal operator of
    basictests.jl:731-754 and rand(300,300) Hermitian (basictests.jl:756-784, where every Lanczos basis loses its orthogonality), and
    H to 1e-11 where the reference's own basis keeps its orthogonality.  profiles/r06_pipelined_lanczos_accuracy.txt is the table."""
    import numpy as np
    import scipy.sparse as sp
    from oracle import pipelined_lanczos as pl
    rng = np.random.default_rng(2026)
    n = 2000
    C2 = sp.diags([np.full(n - abs(o), v) for o, v in zip((-2, -1, 0, 1, 2), (0.3, 1.2, -2.0, 1.2, 0.3))], (-2, -1, 0, 1, 2), format="csr")
    e = np.ones(99)
    P = (-1j * sp.diags([-e, np.zeros(100), e], (-1, 0, 1))).tocsr()
    M = rng.random((300, 300))
    cases = [("c2", C2, rng.standard_normal(n), 1.0, 30, True), ("herm_tridiag", P, rng.random(100) + 1j * rng.random(100), -1.0j, 15, True),
             ("rand300", (M + M.T) / 2, rng.random(300), 1.0, 30, False)]
    for name, A, b, t, m, h_bar in cases:
        ref = pl.lanczos_ref(A, b, m)
        w_ref = pl.expv_from_lanczos(t, *ref, m)
        hmax = max(np.abs(ref[1]).max(), np.abs(ref[2]).max())
        for fn in (pl.lanczos_p1, pl.lanczos_p2, pl.lanczos_p3):
            r = fn(A, b, m)
            w = pl.expv_from_lanczos(t, *r, m)
            assert np.linalg.norm(w - w_ref) <= 1e-12 * np.linalg.norm(w_ref), (name, fn.__name__)
            if h_bar:
                assert max(np.abs(r[1] - ref[1]).max(), np.abs(r[2] - ref[2]).max()) <= 1e-11 * hmax, (name, fn.__name__)

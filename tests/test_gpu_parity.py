"""GPU parity tests: the HIP path (through the C ABI of libexpv_mi.so) against the oracle on the
same seeded inputs.  Tolerances (all fp64):
    device vs oracle, same m:   |H_d - H_o| / |H| <= 1e-12,  |w_d - w_o| / |w_o| <= 1e-12
    vs dense truth exp(tA)b in the converged regime: <= 1e-10 (the reference's own bar is sqrt(eps))
They read like test/basictests.jl and test/gpu/gputests.jl (CPU result vs device result)."""
import numpy as np
import pytest
import scipy.linalg as sl
import scipy.sparse as sp

from oracle import c_oracle as co
from oracle import krylov_oracle as ko
from tests._util import c2_operator, close, dense_phis, mkA, relerr, stencil2d

pytestmark = pytest.mark.gpu
TOL = 1e-12
SQRT_EPS = float(np.sqrt(np.finfo(float).eps))


@pytest.fixture(scope="module")
def eu():
    import expv_mi_loader
    return expv_mi_loader.load()


def herr(Hd, Ho):
    return float(np.max(np.abs(Hd - Ho)) / max(np.max(np.abs(Ho)), 1e-300))


# ------------------------------------------------------------------ Arnoldi / Lanczos --------
@pytest.mark.parametrize("ortho", ["mgs", "lowsync"])
@pytest.mark.parametrize("n,m,iop", [(20, 5, 0), (100, 30, 0), (513, 30, 0), (2000, 30, 0), (2000, 25, 2),
                                     (2001, 17, 3), (4099, 40, 0)])
def test_arnoldi_H_V_parity_sparse_real(eu, n, m, iop, ortho):
    A = c2_operator(n)
    b = np.random.default_rng(3).standard_normal(n)
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, m)
    eu.arnoldi_(Ks, A, b, m=m, iop=iop, ishermitian=False, ortho=ortho)
    Ko = ko.KrylovSubspace(float, float, n, m)
    ko.arnoldi_(Ko, A, b, m=m, iop=iop, ishermitian=False)
    assert Ks.m == Ko.m and Ks.wasbreakdown == Ko.wasbreakdown
    assert abs(Ks.beta - Ko.beta) <= 1e-14 * Ko.beta
    close(Ks.getH(), Ko.getH(), TOL, "arnoldi H sparse real n=%d m=%d iop=%d %s" % (n, m, iop, ortho), mat=True)
    close(Ks.getV(), Ko.getV(), TOL, "arnoldi V sparse real n=%d m=%d iop=%d %s (max abs)" % (n, m, iop, ortho), absolute=True)


@pytest.mark.parametrize("ortho", ["mgs", "lowsync"])
@pytest.mark.parametrize("kind", ["dense_real", "dense_complex", "sparse_complex"])
def test_arnoldi_parity_other_operators(eu, kind, ortho):
    rng = np.random.default_rng(12)
    n, m = 300, 20
    if kind == "dense_real":
        A = rng.standard_normal((n, n)) / np.sqrt(n)
        b = rng.standard_normal(n)
    elif kind == "dense_complex":
        A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    else:
        A = (c2_operator(n) * (1 + 0.25j)).tocsc()
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=False, ortho=ortho)
    Ko = ko.arnoldi(A, b, m=m, ishermitian=False)
    assert Ks.m == Ko.m
    close(Ks.getH(), Ko.getH(), TOL, "arnoldi H %s %s" % (kind, ortho), mat=True)
    close(Ks.getV(), Ko.getV(), TOL, "arnoldi V %s %s (max abs)" % (kind, ortho), absolute=True)


@pytest.mark.parametrize("cplx", [False, True])
def test_lanczos_parity(eu, cplx):
    rng = np.random.default_rng(13)
    n, m = 1500, 30
    if cplx:
        X = rng.standard_normal((200, 200)) + 1j * rng.standard_normal((200, 200))
        A, n = (X + X.conj().T) / 2, 200
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    else:
        A = c2_operator(n, sym=True)
        b = rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m)          # ishermitian(A) -> lanczos!
    Ko = ko.arnoldi(A, b, m=m)
    assert Ks.U == np.float64 and Ks.m == Ko.m
    close(Ks.getH(), Ko.getH(), TOL, "lanczos H cplx=%s" % cplx, mat=True)
    close(Ks.getV(), Ko.getV(), TOL, "lanczos V cplx=%s (max abs)" % cplx, absolute=True)


def test_hermitian_H_real_arnoldi_vs_lanczos(eu):
    """basictests.jl:731-754 on the device."""
    rng = np.random.default_rng(7)
    n, m = 100, 15
    e = np.ones(n)
    p = -1j * (np.diag(-e[1:], -1) + np.diag(e[1:], 1))
    v = rng.random(n) + 1j * rng.random(n)
    KsA = eu.KrylovSubspace(np.complex128, np.complex128, n, m)
    KsL = eu.KrylovSubspace(np.complex128, np.float64, n, m)
    eu.arnoldi_(KsA, p, v, ishermitian=False)
    eu.lanczos_(KsL, p, v)
    AH = KsA.H[: KsA.m, : KsA.m]
    LH = KsL.H[: KsL.m, : KsL.m]
    assert np.linalg.norm(AH - LH) / np.linalg.norm(AH) < 1e-14


def test_happy_breakdown_and_zero_input(eu):
    """basictests.jl:544-553, :565-566."""
    rng = np.random.default_rng(5)
    n = 20
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    A = np.outer(v, v)
    b = rng.standard_normal(n)
    assert eu.arnoldi(A, b).m == 2
    Ks = eu.arnoldi(A, b, ishermitian=False)
    assert Ks.m == 2 and Ks.wasbreakdown
    for herm in (False, True):
        M = rng.standard_normal((n, n))
        if herm:
            M = (M + M.T) / 2
        wz = eu.expv(1e-2, M, np.zeros(n), m=5)
        assert np.linalg.norm(wz) == 0.0 and not np.any(np.isnan(wz))


def test_dimension_mismatch(eu):
    A = c2_operator(30)
    Ks = eu.KrylovSubspace(np.float64, np.float64, 31, 5)
    with pytest.raises(eu.DimensionMismatch):
        eu.arnoldi_(Ks, A, np.ones(30))
    with pytest.raises(eu.DimensionMismatch):
        eu.arnoldi(A, np.ones(29))


def test_resize_and_continuation(eu):
    """arnoldi!(...; init=j) continues in place (arnoldi.jl:350,360-368)."""
    n = 400
    A = c2_operator(n)
    b = np.random.default_rng(3).standard_normal(n)
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, 20)
    eu.arnoldi_(Ks, A, b, m=8, ishermitian=False)
    eu.arnoldi_(Ks, A, b, m=20, init=8, ishermitian=False)
    Ko = ko.KrylovSubspace(float, float, n, 20)
    ko.arnoldi_(Ko, A, b, m=20, ishermitian=False)
    assert herr(Ks.getH(), Ko.getH()) <= TOL
    Ks.resize(25)
    assert Ks.maxiter == 25 and Ks.m == 25 and np.all(Ks.H == 0)


# ------------------------------------------------------------------ expv / phiv --------------
@pytest.mark.parametrize("kindA", ["hc", "hr", "gc", "gr"])
@pytest.mark.parametrize("cb", [True, False])
@pytest.mark.parametrize("t", [1e-2, 1e-2j, 1e-2 + 1e-2j])
def test_complex_value_matrix(eu, kindA, cb, t):
    """basictests.jl:650-664 on the device, plus parity with the oracle."""
    rng = np.random.default_rng(abs(hash((kindA, cb))) % 1000)
    n, m = 20, 10
    X = rng.random((n, n)) + (1j * rng.random((n, n)) if kindA[1] == "c" else 0)
    A = (X + X.conj().T) / 2 if kindA[0] == "h" else X
    b = rng.random(n) + (1j * rng.random(n) if cb else 0)
    w = eu.expv(t, A, b, m=m)
    assert relerr(w, sl.expm(t * A) @ b) < SQRT_EPS
    assert relerr(w, ko.expv(t, A, b, m=m)) < TOL


def test_arnoldi_krylov_testset(eu):
    """basictests.jl:515-541."""
    rng = np.random.default_rng(0)
    n, m, K, t = 20, 5, 4, 1e-2
    A = rng.standard_normal((n, n))
    b = rng.standard_normal(n)
    direct = sl.expm(t * A) @ b
    assert relerr(eu.expv(t, A, b, m=m), direct) < SQRT_EPS
    w, stats = eu.kiops(t, A, b)
    assert relerr(w[:, 0], direct) < SQRT_EPS
    P = dense_phis(t * A, K)
    W = np.stack([P[i] @ b for i in range(K + 1)], axis=1)
    Ks = eu.arnoldi(A, b, m=m)
    assert relerr(eu.phiv(t, Ks, K), W) < SQRT_EPS
    w3, st3 = eu.kiops(t, A, np.stack([b * (1 / t) ** i for i in range(K)], axis=1))
    assert relerr(w3[:, 0], W[:, :K].sum(axis=1)) < SQRT_EPS
    wo, so = ko.kiops(t, A, np.stack([b * (1 / t) ** i for i in range(K)], axis=1))
    assert st3 == so
    # the one bar above 1e-12 in this file: the columns b * (1/t)^i span six orders of magnitude (t = 1e-2), kiops rescales the
    # augmented block by mu = 2^20 (kiops.jl:94-103) and the result (norm ~ 4 |b|) is what is left after cancelling terms of
    # size 1e6 |b|: two correct evaluations differ by ~1e6 eps.  Measured on MI355X: 7.5e-11 (profiles/r02_parity_measured.txt)
    close(w3, wo, 1e-9, "kiops n=20 dense, 4 columns spanning 1e6 in scale vs oracle")


def test_phiv_matrix_kat(eu):
    """basictests.jl:569-573 (m = n => exact)."""
    n = 30
    A = np.diag(np.ones(n - 1), -1) + 30 * np.eye(n) + np.diag(np.ones(n - 1), 1)
    t = 0.1
    Q = eu.phiv(t, A, np.ones(n), 10)
    ref = np.linalg.solve(t * A, (sl.expm(t * A) - np.eye(n)) @ np.ones(n))
    assert relerr(Q[:, 1], ref) < SQRT_EPS
    np.testing.assert_allclose(Q[:3, 1], [6.85734928, 7.33460365, 7.3533841], rtol=1e-8)


@pytest.mark.parametrize("correct", [False, True])
def test_phiv_parity_and_errest(eu, correct):
    n, m, k = 64, 30, 3
    A = mkA(n)
    b = 1.0 / np.arange(1, n + 1)
    Ks = eu.arnoldi(A, b, m=m)
    Ko = ko.arnoldi(A, b, m=m)
    W, err = eu.phiv_(np.empty((n, k + 1), order="F"), 0.1, Ks, k, correct=correct, errest=True)
    Wo, erro = ko.phiv_(np.empty((n, k + 1), order="F"), 0.1, Ko, k, correct=correct, errest=True)
    assert relerr(W, Wo) < TOL
    assert abs(err - erro) <= 1e-9 * max(erro, 1e-300) + 1e-25


@pytest.mark.parametrize("herm", [False, True])
def test_matrix_free_operator(eu, herm):
    """basictests.jl:786-816: an operator with only eltype/size/mul!/ishermitian (device callback)."""
    import torch
    rng = np.random.default_rng(123)
    n = 20
    A = rng.random((n, n)) + 1j * rng.random((n, n))
    M = A.conj().T @ A if herm else A
    Md = torch.as_tensor(M, device="cuda")
    Op = eu.MIOperator(None, matvec=lambda x: Md @ x, shape=(n, n), dtype=np.complex128, ishermitian=herm)
    b = rng.random(n) + 1j * rng.random(n)
    Ks = eu.arnoldi(Op, b, ishermitian=herm, tol=1e-12)
    pv = eu.phiv(0.01, Ks, 2)
    ref = np.stack([P @ b for P in dense_phis(0.01 * M, 2)], axis=1)
    np.testing.assert_allclose(pv, ref, atol=1e-12, rtol=SQRT_EPS)
    np.testing.assert_allclose(eu.expv(0.01, Op, b, m=n, ishermitian=herm), sl.expm(0.01 * M) @ b, atol=1e-12,
                               rtol=SQRT_EPS)


def test_error_estimate_mode(eu):
    """basictests.jl:756-784."""
    rng = np.random.default_rng(9)
    n, m, dt = 300, 30, 0.1
    A = rng.random((n, n))
    A = (A + A.T) / 2
    b = rng.random(n) + 1j * rng.random(n)
    w = eu.expv(-1j, dt * A, b, m=m, tol=1e-10, rtol=1e-10, mode="error_estimate")
    wp = sl.expm(-1j * dt * A) @ b
    dw = np.linalg.norm(w - wp)
    assert dw < 1e-10 and dw / abs(1e-16 + np.linalg.norm(w)) < 1e-10
    wo = ko.expv(-1j, dt * A, b, m=m, tol=1e-10, rtol=1e-10, mode="error_estimate")
    # rand(n, n) has one dominant eigenvalue (~n/2 against |lambda| < ~5): its Ritz value converges after ~5 steps and
    # the Lanczos basis then loses orthogonality completely (max |V'V - I| = 0.79 at the stopping step m = 11, in the
    # reference's recurrence too), so two correct recurrences differ by rounding x that amplification: measured 6.6e-12
    close(w, wo, 1e-10, "error_estimate mode vs oracle (basis has lost orthogonality: 0.79)")
    wz = eu.expv(-1j, dt * A, np.zeros(n, dtype=complex), m=m, tol=1e-10, rtol=1e-10, mode="error_estimate")
    assert np.linalg.norm(wz) == 0
    with pytest.raises(RuntimeError):
        eu.expv(-1j, rng.random((5, 5)), np.ones(5, dtype=complex), mode="error_estimate", ishermitian=False)


# ------------------------------------------------------------------ time stepping ------------
def test_issue_143(eu):
    """basictests.jl:193-205: 1x1 operator, breakdown => one step, stdout line."""
    ts = np.arange(0, 1.0001, 0.1)
    out = []
    res = eu.expv_timestep(ts.copy(), np.array([[1.0]]), np.array([1.0]), verbose=True, out=out.append)
    assert any("Completed after 1 time step(s)" in s for s in out)
    np.testing.assert_allclose(np.asarray(res).ravel(), np.exp(ts), rtol=SQRT_EPS)


def test_adaptive_krylov(eu):
    """basictests.jl:666-691 and control-flow parity with the oracle (same steps, same matvecs)."""
    n, K, t, tol = 100, 4, 5.0, 1e-7
    A = sp.diags([np.ones(n - 1), -2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csc")
    B = np.random.default_rng(14).standard_normal((n, K + 1))
    Ad = A.toarray()
    Ph, Phh = dense_phis(t * Ad, K), dense_phis(t / 2 * Ad, K)
    u_exact = sum(t ** i * Ph[i] @ B[:, i] for i in range(K + 1))
    uhalf = sum((t / 2) ** i * Phh[i] @ B[:, i] for i in range(K + 1))
    st, so = {}, {}
    U = eu.phiv_timestep(np.array([t / 2, t]), A, B, adaptive=True, tol=tol, stats=st)
    Uo = ko.phiv_timestep(np.array([t / 2, t]), A, B, adaptive=True, tol=tol, stats=so)
    assert relerr(U[:, 0], uhalf) < tol and relerr(U[:, 1], u_exact) < tol
    assert st["num_timesteps"] == so["num_timesteps"] and st["matvecs"] == so["matvecs"] and st["m"] == so["m"]
    close(U, Uo, TOL, "phiv_timestep adaptive sparse n=100 U vs oracle")
    u_exact0 = Ph[0] @ B[:, 0]
    opn = lambda M, p: abs(M).sum(axis=1).max()
    assert relerr(eu.expv_timestep(t, A, B[:, 0], adaptive=True, tol=tol, opnorm=opn), u_exact0) < tol
    assert relerr(eu.expv_timestep(t, A, B[:, 0], adaptive=True, tol=tol, opnorm=opn(A, np.inf)), u_exact0) < tol


def test_matrix_free_default_tolerance(eu):
    """basictests.jl:693-729."""
    n, t, tol = 50, 3.0, 1e-7
    Ad = sp.diags([np.ones(n - 1), -2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csc")
    b = np.random.default_rng(8).standard_normal(n)
    u_exact = sl.expm(t * Ad.toarray()) @ b
    assert relerr(eu.expv_timestep(t, Ad, b, adaptive=True, tol=tol), u_exact) < 1e-5
    assert relerr(eu.expv_timestep(t, Ad, b, adaptive=True, tol=tol, opnorm=4.0), u_exact) < 1e-5


def test_expv_timestep_300_snapshots_complex_sparse(eu):
    """test/gpu/gputests.jl:41-58: n=1000 complex upper-triangular-ish sparse, 300 snapshots, CPU vs device."""
    rng = np.random.default_rng(0x0451)
    n = 1000
    A = sp.random(n, n, density=10 / n, random_state=rng, dtype=np.float64) \
        + 1j * sp.random(n, n, density=10 / n, random_state=rng, dtype=np.float64)
    A = (sp.triu(A, 1) + sp.random(n, n, density=1 / n, random_state=rng) * (1 + 1j)).tocsc()
    b = rng.random(n) + 1j * rng.random(n)
    t = 0.1
    assert relerr(eu.expv(t, A, b), ko.expv(t, A, b)) < TOL
    ts = np.linspace(0, 1, 300)
    E1 = ko.expv_timestep(ts.copy(), A, b)
    E2 = eu.expv_timestep(ts.copy(), A, b)
    # (the reference's own bar for this test is sqrt(eps): test/gpu/gputests.jl:58 uses isapprox)
    close(E2, E1, TOL, "expv_timestep 300 snapshots complex sparse vs oracle")


def test_kiops_parity(eu):
    rng = np.random.default_rng(11)
    n = 400
    A = c2_operator(n).tocsc()
    u = rng.standard_normal((n, 3))
    for herm_A in (A, c2_operator(n, sym=True).tocsc()):
        w, st = eu.kiops(1.0, herm_A, u)
        wo, so = ko.kiops(1.0, herm_A, u)
        assert st == so
        close(w, wo, TOL, "kiops real n=400, 3 columns vs oracle")
    with pytest.raises(eu.DimensionMismatch):
        eu.kiops(np.array([[0.5, 1.0]]), A, u[:, 0])
    with pytest.raises(TypeError):
        eu.kiops(1.0, A.astype(complex), u[:, 0])


def test_kiops_complex_extension(eu):
    """BASELINE config 4 shape at small n: no reference behaviour ("parity unpinned"); pinned by the
    dense expm of the operator and by the oracle's same extension."""
    rng = np.random.default_rng(6)
    n = 300
    A = (c2_operator(n) * (1 + 0.25j)).tocsc()
    u = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    w, st = eu.kiops(1.0, A, u, allow_complex=True, ishermitian=False)
    truth = sl.expm(A.toarray()) @ u
    assert relerr(w[:, 0], truth) < 1e-6
    wo, so = ko.kiops(1.0, A, u, allow_complex=True, ishermitian=False)
    assert st == so
    close(w, wo, TOL, "kiops complex extension n=300 vs oracle")


# ------------------------------------------------------------------ full-size properties -----
def test_c2_full_size_parity_with_c_oracle(eu):
    """BASELINE config 2 at full size (n = 1e6, m = 30): H and w against the plain-C restatement."""
    n, m = 1_000_000, 30
    A = c2_operator(n)
    b = np.random.default_rng(3).standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m)
    r = co.arnoldi_csr(A, b, m=m)
    assert Ks.m == r["m"] == m
    close(Ks.getH(), r["H"], TOL, "C2 full size H vs C oracle", mat=True)
    w = eu.expv_(np.empty(n), 1.0, Ks)
    wo, _ = co.expv_csr(1.0, A, b, m=m)
    close(w, wo, TOL, "C2 full size w vs C oracle")
    # size-independent properties: group property and linearity of exp(tA)
    back = eu.expv(-1.0, A, w, m=m)
    close(back, b, 1e-9, "C2 full size group property exp(-A) exp(A) b = b (m = 30 truncation both ways)")
    w2 = eu.expv(1.0, A, 2.5 * b, m=m)
    close(w2, 2.5 * w, 1e-13, "C2 full size linearity")


def test_c2_symmetric_and_stencil_variants(eu):
    n, m = 200_000, 30
    b = np.random.default_rng(3).standard_normal(n)
    As = c2_operator(n, sym=True)
    w = eu.expv(1.0, As, b, m=m)
    wo, r = co.expv_csr(1.0, As, b, m=m, hermitian=True)
    close(w, wo, TOL, "C2 symmetric variant n=2e5 (Lanczos) w vs C oracle")
    St = stencil2d(448)
    bs = np.random.default_rng(4).standard_normal(St.shape[0])
    ws = eu.expv(0.2, St, bs, m=m)
    wso, _ = co.expv_csr(0.2, St, bs, m=m)
    close(ws, wso, TOL, "2-D stencil n=448^2 w vs C oracle")


def test_device_resident_vectors(eu):
    """Inputs already in HBM (torch tensors): nothing is staged through the host."""
    import torch
    n, m = 50_000, 30
    A = c2_operator(n)
    b = np.random.default_rng(3).standard_normal(n)
    bd = torch.as_tensor(b, device="cuda")
    wd = eu.expv(1.0, A, bd, m=m)
    assert wd.is_cuda
    assert relerr(wd.cpu().numpy(), ko.expv(1.0, A, b, m=m, ishermitian=False)) < TOL


# ------------------------------------------------------------------ batch (config 5) ---------
@pytest.mark.parametrize("sym", [False, True])
def test_expv_batch_matches_single_problem_loop(eu, sym):
    """BASELINE config 5 at small size: nprob operators with one pattern, values scaled per problem."""
    rng = np.random.default_rng(7)
    n, nprob, m = 3000, 9, 30
    A0 = c2_operator(n, sym=sym).tocsr()
    A0.sort_indices()
    scales = 1 + 0.1 * rng.random(nprob)
    vals = np.stack([A0.data * s for s in scales])
    B = np.asfortranarray(rng.standard_normal((n, nprob)))
    ts = np.linspace(0.5, 1.0, nprob)
    W, mu = eu.expv_batch(ts, A0, vals, B, m=m, ishermitian=sym, return_m=True)
    for p in range(nprob):
        Ap = A0.copy()
        Ap.data = vals[p].copy()
        wo = ko.expv(ts[p], Ap, B[:, p], m=m, ishermitian=sym)
        assert relerr(W[:, p], wo) < TOL
        assert mu[p] == m


@pytest.mark.parametrize("T", [np.float32, np.complex64, np.complex128])
@pytest.mark.parametrize("kind", ["banded", "general"])
def test_expv_batch_other_element_types(eu, T, kind):
    """the batched step for every BlasFloat (ExponentialUtilities.jl:19) -- Float32 batches on the batched
    single-pass step (32-bit storage, tiles of 1024 rows), everything else through the batched two-kernel step -- against a
    loop of the oracle's expv over the problems, at the bars of the element type; sizes off the tile boundaries, one problem
    with a zero right-hand side."""
    rng = np.random.default_rng(23)
    cplx = np.dtype(T).kind == "c"
    n, nprob, m = 2500, 5, 16
    if kind == "banded":
        A0 = (c2_operator(n) * ((1 + 0.25j) if cplx else 1.0)).tocsr()
    else:
        rows = np.repeat(np.arange(n), 3)
        A0 = (sp.coo_matrix((rng.standard_normal(3 * n) * 0.3, (rows, rng.integers(0, n, size=3 * n))), shape=(n, n)).tocsr()
              + sp.diags([np.full(n, -0.5)], [0], format="csr")).tocsr()
        A0.sum_duplicates()
        if cplx:
            A0 = (A0 * (1 + 0.25j)).tocsr()
    A0.sort_indices()
    A0 = A0.astype(T)
    vals = np.stack([A0.data * s for s in (1 + 0.1 * rng.random(nprob))]).astype(T)
    B = np.asfortranarray((rng.standard_normal((n, nprob)) + (1j * rng.standard_normal((n, nprob)) if cplx else 0)).astype(T))
    B[:, 3] = 0
    T64 = np.complex128 if cplx else np.float64
    tol = 3e-5 if np.dtype(T).itemsize <= 8 and T != np.float64 and T != np.complex128 else TOL
    W, mu = eu.expv_batch(0.8, A0, vals, B, m=m, return_m=True)
    assert np.asarray(W).dtype == np.dtype(T)
    for p in range(nprob):
        Ap = A0.astype(T64).copy()
        Ap.data = vals[p].astype(T64)
        wo = ko.expv(0.8, Ap, B[:, p].astype(T64), m=m, ishermitian=False)
        if p == 3:
            assert np.all(np.asarray(W)[:, p] == 0)
        else:
            close(np.asarray(W)[:, p].astype(T64), wo, tol, "expv_batch %s %s: problem %d vs the oracle's expv" % (kind, np.dtype(T).name, p))


def test_expv_batch_breakdown_and_zero_columns(eu):
    """Per-problem state: one problem breaks down early, one has a zero right-hand side."""
    n, m = 256, 20
    A0 = sp.identity(n, format="csr") * 2.0          # A = 2I: Krylov space has dimension 1 -> breakdown at j = 1
    vals = np.stack([A0.data, A0.data * 0.5, A0.data])
    rng = np.random.default_rng(1)
    B = np.asfortranarray(rng.standard_normal((n, 3)))
    B[:, 2] = 0.0
    W, mu = eu.expv_batch(0.3, A0, vals, B, m=m, return_m=True)
    np.testing.assert_allclose(W[:, 0], np.exp(0.6) * B[:, 0], rtol=1e-13)
    np.testing.assert_allclose(W[:, 1], np.exp(0.3) * B[:, 1], rtol=1e-13)
    assert np.all(W[:, 2] == 0.0)
    assert mu[0] == 1 and mu[1] == 1


# ------------------------------------------------------------------ edge cases ---------------
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 1023, 1025])
def test_sizes_around_tile_boundaries(eu, n):
    """Ragged sizes: one row, odd lengths (16-byte packs), wave (128) and tile boundaries."""
    rng = np.random.default_rng(n)
    A = c2_operator(n) if n > 2 else sp.csr_matrix(np.array([[-2.0]]) if n == 1 else np.array([[-2.0, 0.8], [1.2, -2.0]]))
    b = rng.standard_normal(n)
    m = min(30, n)
    for ortho in ("auto", "mgs"):
        Ks = eu.arnoldi(A, b, m=m, ishermitian=False, ortho=ortho)
        Ko = ko.arnoldi(A, b, m=m, ishermitian=False)
        assert Ks.m == Ko.m and Ks.wasbreakdown == Ko.wasbreakdown
        mm = Ks.m
        close(Ks.H[: mm + 1, :mm], Ko.H[: mm + 1, :mm], TOL, "H at ragged size n=%d %s" % (n, ortho), mat=True)
        w = eu.expv_(np.empty(n), 0.7, Ks)
        close(w, sl.expm(0.7 * A.toarray()) @ b, 1e-9, "expv at ragged size n=%d vs dense truth (m = 30 truncation)" % n)


def test_m_larger_than_n_breaks_down(eu):
    n = 4
    A = c2_operator(8)[:n, :n].tocsr()
    b = np.random.default_rng(2).standard_normal(n)
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, 8)
    eu.arnoldi_(Ks, A, b, m=8, ishermitian=False)
    Ko = ko.KrylovSubspace(float, float, n, 8)
    ko.arnoldi_(Ko, A, b, m=8, ishermitian=False)
    assert Ks.m == Ko.m <= n and Ks.wasbreakdown and Ko.wasbreakdown


def powerlaw_matrix(n, seed, cplx=False, local=0, longest=None):
    """irregular rows: Zipf-distributed lengths (mean ~5, a few rows with hundreds of entries), random columns (anywhere, or
    within +-local of the row); test/gpu/gputests.jl:41-58 uses sprand -- this is the harder, skewed version of it"""
    rng = np.random.default_rng(seed)
    ln = np.minimum(rng.zipf(1.8, size=n), n // 2)
    ln = np.maximum(1, (ln * (5.0 / ln.mean())).astype(np.int64))
    ln = np.minimum(ln, n - 1)
    if longest is not None:
        ln[n // 3] = longest                                   # one row long enough for several overflow segments
    rows = np.repeat(np.arange(n), ln)
    cols = rng.integers(0, n, size=rows.size) if not local else np.clip(rows + rng.integers(-local, local + 1, size=rows.size), 0, n - 1)
    vals = rng.standard_normal(rows.size) / np.sqrt(np.repeat(ln, ln))
    if cplx:
        vals = vals * (1 + 0.3j) + 0.1j * rng.standard_normal(rows.size) / np.sqrt(np.repeat(ln, ln))
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr() + sp.diags([np.full(n, -0.5)], [0], format="csr")   # (a larger shift makes MGS itself lose orthogonality: y = A v ~ shift * v)
    A.sum_duplicates()
    A[7, :] = 0                                                # an empty row
    A.eliminate_zeros()
    return A.tocsr()


@pytest.mark.parametrize("case", ["real", "complex", "real_local", "real_very_long_row", "real_csc_input"])
def test_irregular_rows_sell_cut_plus_overflow(eu, case):
    """Power-law row lengths: SELL slots up to a cut-off + the overflow pass (segments of the CSR arrays, 8 lanes per segment,
    several segments for a very long row), on the two-kernel step (full Arnoldi, IOP, Lanczos is not applicable) and on the
    modular launches (strict MGS), against the oracle; also through expv / phiv and mul!."""
    n, m = 6000, 30
    cplx = case == "complex"
    A = powerlaw_matrix(n, 77, cplx=cplx, local=400 if case == "real_local" else 0, longest=1500 if case == "real_very_long_row" else None)
    if case == "real_csc_input":
        A = A.tocsc()
    info = eu.host_pattern_info(A.tocsr(), np.complex128 if cplx else np.float64)
    assert info["sell"] and info["sell_cut"] > 0 and "overflow" in info["path"], info
    rng = np.random.default_rng(5)
    b = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    op = eu.MIOperator(A)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    close(op @ x, A @ x, 1e-13, "mul! irregular rows %s" % case)
    for ortho, iop in (("lowsync", 0), ("mgs", 0), ("lowsync", 3)):
        Ks = eu.arnoldi(op, b, m=m, ishermitian=False, ortho=ortho, iop=iop)
        Ko = ko.arnoldi(A, b, m=m, ishermitian=False, iop=iop)
        assert Ks.m == Ko.m
        close(Ks.getH(), Ko.getH(), TOL, "arnoldi H irregular rows %s %s iop=%d" % (case, ortho, iop), mat=True)
        close(Ks.getV(), Ko.getV(), TOL, "arnoldi V irregular rows %s %s iop=%d (max abs)" % (case, ortho, iop), absolute=True)
    w = eu.expv(0.8, op, b, m=m, ishermitian=False)
    assert "two_kernel" in eu.expv.last_stats["path"], eu.expv.last_stats
    close(w, ko.expv(0.8, A, b, m=m, ishermitian=False), TOL, "expv irregular rows %s" % case)
    close(eu.phiv(0.5, op, b, 2, m=20), ko.phiv(0.5, A, b, 2, m=20), 1e-11, "phiv irregular rows %s" % case)
    if not cplx:
        # values-only refresh on the same pattern: the SELL slots AND the overflow entries follow
        A2 = A.copy()
        A2.data = A2.data * (1.0 + 0.1 * np.cos(np.arange(A2.nnz)))
        op.update_values(A2)
        close(eu.expv(0.8, op, b, m=m, ishermitian=False), ko.expv(0.8, A2.tocsr(), b, m=m, ishermitian=False), TOL,
              "expv irregular rows %s after update_values" % case)


def test_irregular_rows_kiops_and_timestep(eu):
    """the augmented operator of kiops and the adaptive phiv_timestep on an irregular-row operator (two-kernel step + overflow pass)"""
    n = 4000
    A = powerlaw_matrix(n, 91, local=300) * 0.5
    rng = np.random.default_rng(8)
    u = rng.standard_normal((n, 3))
    w, st = eu.kiops(0.7, A, u)
    wo, so = ko.kiops(0.7, A, u)
    assert tuple(st) == tuple(so)
    close(w, wo, 1e-10, "kiops on an irregular-row operator")
    B = rng.standard_normal((n, 3))
    s1, s2 = {}, {}
    U = eu.phiv_timestep(np.array([0.4, 1.0]), A, B, adaptive=True, tol=1e-8, stats=s1)
    Uo = ko.phiv_timestep(np.array([0.4, 1.0]), A, B, adaptive=True, tol=1e-8, stats=s2)
    assert (s1["num_timesteps"], s1["matvecs"], s1["m"]) == (s2["num_timesteps"], s2["matvecs"], s2["m"])
    close(U, Uo, 1e-11, "phiv_timestep on an irregular-row operator")


def test_irregular_rows_fall_back_to_csr_and_empty_rows(eu):
    """A dense row makes SELL padding explode (SELL slots up to a cut-off + overflow pass); empty rows are legal."""
    rng = np.random.default_rng(21)
    n = 700
    A = sp.random(n, n, density=0.01, random_state=rng, format="lil")
    A[5, :] = rng.standard_normal(n)          # one dense row
    A[17, :] = 0                              # empty rows
    A[n - 1, :] = 0
    A = (A.tocsr() * 0.3).tocsc()
    b = rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=25, ishermitian=False)
    Ko = ko.arnoldi(A, b, m=25, ishermitian=False)
    assert herr(Ks.getH(), Ko.getH()) <= TOL
    assert relerr(eu.expv(1.0, A, b, m=25), ko.expv(1.0, A, b, m=25)) < TOL


def test_sparse_complex_fused_full_arnoldi(eu):
    rng = np.random.default_rng(31)
    n, m = 5000, 30
    A = (c2_operator(n) * (1 + 0.25j)).tocsc()
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=False)
    Ko = ko.arnoldi(A, b, m=m, ishermitian=False)
    assert herr(Ks.getH(), Ko.getH()) <= TOL
    for t in (0.4, 0.4j, -0.2 + 0.1j):
        w = eu.expv_(np.empty(n, dtype=complex), t, Ks)
        assert relerr(w, ko.expv_(np.empty(n, dtype=complex), t, Ko)) < TOL


def test_window_longer_than_one_chunk_and_iop(eu):
    """m = 48 (> 32): three projection chunks; iop windows of 5 and 20."""
    n = 3000
    A = stencil2d(55)[:n, :n].tocsr()
    b = np.random.default_rng(4).standard_normal(n)
    for m, iop in ((48, 0), (40, 5), (40, 20)):
        Ks = eu.KrylovSubspace(np.float64, np.float64, n, m)
        eu.arnoldi_(Ks, A, b, m=m, iop=iop, ishermitian=False)
        Ko = ko.KrylovSubspace(float, float, n, m)
        ko.arnoldi_(Ko, A, b, m=m, iop=iop, ishermitian=False)
        assert Ks.m == Ko.m
        close(Ks.getH(), Ko.getH(), TOL, "H stencil n=3000 m=%d iop=%d" % (m, iop), mat=True)


def test_timestep_sorts_ts_in_place_and_matrix_output(eu):
    """krylov_phiv_adaptive.jl:297: ts is sorted in place; U[:, j] follows the sorted order."""
    n = 60
    A = c2_operator(n).tocsc()
    b = np.random.default_rng(5).standard_normal(n)
    ts = np.array([0.9, 0.1, 0.5])
    U = eu.expv_timestep(ts, A, b, tol=1e-9)
    Uo = ko.expv_timestep(np.array([0.9, 0.1, 0.5]), A, b, tol=1e-9)
    close(U, Uo, TOL, "expv_timestep unsorted ts vs oracle")
    for j, t in enumerate(sorted([0.9, 0.1, 0.5])):       # non-adaptive, m = 10: the method's own accuracy
        assert relerr(U[:, j], sl.expm(t * A.toarray()) @ b) < 1e-5


def test_pipelined_path_when_enabled_is_lazy_and_consistent(eu):
    """Ks.V is orthonormal whenever it is observed, whatever path produced it."""
    n, m = 4000, 20
    A = c2_operator(n)
    b = np.random.default_rng(6).standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=False)
    w1 = eu.expv_(np.empty(n), 1.0, Ks)          # combine before V is observed
    V = Ks.getV()
    G = V.T @ V
    assert np.max(np.abs(G - np.eye(m + 1))) < 1e-12
    w2 = eu.expv_(np.empty(n), 1.0, Ks)          # and after
    assert relerr(w2, w1) < 1e-14


@pytest.mark.parametrize("case", ["gappy_diagonals", "nine_offsets", "dropped_entries", "lanczos_tridiag"])
def test_banded_operator_forms(eu, case):
    """Narrow-banded operators take the single-pass pipeline; their diagonals are read in DIA form when there are
    at most 8 distinct offsets and little zero fill, in SELL form otherwise.  Same H, V and expv either way."""
    rng = np.random.default_rng(11)
    n, m = 3001, 24
    herm = False
    if case == "gappy_diagonals":        # offsets with holes, all diagonals full
        offs = [-5, -1, 0, 3, 8]
    elif case == "nine_offsets":         # more distinct offsets than the DIA form takes -> SELL slots
        offs = [-4, -3, -2, -1, 0, 1, 2, 3, 4]
    elif case == "dropped_entries":      # ~10 % of the entries absent: explicit zeros in the DIA form
        offs = [-2, -1, 0, 1, 2]
    else:
        offs = [-1, 0, 1]
        herm = True
    diags = [rng.standard_normal(n - abs(o)) * 0.4 - (2.0 if o == 0 else 0.0) for o in offs]
    A = sp.diags(diags, offs, shape=(n, n), format="csr")
    if case == "dropped_entries":
        A = A.tocoo()
        keep = rng.random(A.nnz) > 0.1
        A = sp.csr_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=(n, n))
    if herm:
        A = ((A + A.T) * 0.5).tocsr()
    b = rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=herm)
    Ko = ko.arnoldi(A, b, m=m, ishermitian=herm)
    assert Ks.m == Ko.m and Ks.wasbreakdown == Ko.wasbreakdown
    # the random bands make the basis lose orthogonality (the reference's MGS does too): rounding differences
    # between two correct orthogonalisations are amplified by that loss, so the bar scales with it
    Vo = Ko.V[:, : m + 1]
    loss = float(np.max(np.abs(Vo.T @ Vo - np.eye(m + 1))))
    print("[parity] %-90s loss of orthogonality of the oracle's own basis: %.3e" % ("banded forms: " + case, loss))
    close(Ks.H[: m + 1, :m], Ko.H[: m + 1, :m], max(TOL, 10 * loss), "banded form %s: H" % case, mat=True)
    w = eu.expv_(np.empty(n), 0.5, Ks)
    wo = ko.expv_(np.empty(n), 0.5, Ko)
    close(w, wo, max(TOL, 10 * loss), "banded form %s: w" % case)
    close(Ks.getV(), Ko.getV(), max(TOL, 100 * loss), "banded form %s: V (max abs)" % case, absolute=True)
    close(eu.expv(0.5, A, b, m=m, ishermitian=herm), wo, max(TOL, 10 * loss), "banded form %s: whole-call w" % case)
    # FIXED bars next to the scaled ones: the Arnoldi relation A V_m = V_{m+1} H holds to rounding for ANY correct
    # orthogonalisation, whatever orthogonality the basis has lost -- residual relative to |A| |V| at 1e-13 for the device basis
    # (the oracle's own residual is printed beside it), and beta / the first column at 1e-14
    Vd, Hd = Ks.getV(), Ks.H[: m + 1, :m]
    Ad = A.toarray() if hasattr(A, "toarray") else np.asarray(A)
    scale = float(np.linalg.norm(Ad, 2))
    res_d = float(np.max(np.abs(Ad @ Vd[:, :m] - Vd @ Hd)) / scale)
    res_o = float(np.max(np.abs(Ad @ Vo[:, :m] - Vo @ Ko.H[: m + 1, :m])) / scale)
    print("[parity] %-90s Arnoldi relation residual: device %.3e, oracle %.3e" % ("banded forms: " + case, res_d, res_o))
    close(res_d, 0.0, 1e-13, "banded form %s: |A V_m - V_{m+1} H| / |A| (fixed bar)" % case, absolute=True)
    close(Ks.beta, Ko.beta, 1e-14 * Ko.beta, "banded form %s: beta (fixed bar)" % case, absolute=True)
    close(Vd[:, 0], Vo[:, 0], 1e-14, "banded form %s: v_1 (max abs, fixed bar)" % case, absolute=True)


def test_pipeline_overlap_options_and_expired_wait_fallback(eu):
    """The overlapped pipeline, the one-launch-after-the-other form and the redo after an expired wait (forced
    here by a spin limit of one poll, in a subprocess because the limit is read once) give the same result."""
    import subprocess, sys, os, textwrap
    n, m = 20000, 20
    A = c2_operator(n)
    b = np.random.default_rng(8).standard_normal(n)
    wo = ko.expv(0.9, A, b, m=m, ishermitian=False)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    w1 = eu.expv(0.9, op, b, m=m, ishermitian=False)
    ctx.set_pipeline_overlap(False)
    w2 = eu.expv(0.9, op, b, m=m, ishermitian=False)
    ctx.set_pipeline_overlap(True)
    assert relerr(w1, wo) < 1e-12 and relerr(w2, wo) < 1e-12
    code = textwrap.dedent("""
        import sys, numpy as np, scipy.sparse as sp
        sys.path.insert(0, %r)
        import expv_mi_loader
        from tests._util import c2_operator
        eu = expv_mi_loader.load()
        A = c2_operator(%d)
        G = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-150, -1, 0, 1, 150], shape=A.shape, format="csr")   # wave form
        b = np.random.default_rng(8).standard_normal(%d)
        for _ in range(3):
            w = eu.expv(0.9, A, b, m=%d, ishermitian=False)
            g = eu.expv(0.9, G, b, m=%d, ishermitian=False)
        np.save(sys.argv[1], np.stack([w, g]))
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, n, m, m)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "expv_mi_fallback_%d.npy" % os.getpid())
    G = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-150, -1, 0, 1, 150], shape=A.shape, format="csr")
    wg = ko.expv(0.9, G, b, m=m, ishermitian=False)
    for patch in ("0", "1"):      # the grid in its natural ordering (wave form -> two-kernel step) and in the grid-patch ordering (patch form -> serial redo)
        env = dict(os.environ, EXPV_MI_PIPE_SPIN_LIMIT="1", EXPV_MI_PATCH=patch)
        r = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.load(out)
        assert relerr(got[0], wo) < 1e-12
        close(got[1], wg, TOL, "grid stencil after the expired wait (EXPV_MI_PATCH=%s)" % patch)
    os.remove(out)


def test_overlapped_and_serial_pipeline_are_bitwise_identical(eu):
    """Same kernels, same arithmetic, same reduction tree: only the hand-over between steps differs (flags and
    write-through memory traffic instead of kernel boundaries), so H, V and w must agree to the last bit --
    over ragged sizes, short and long windows, IOP, Lanczos and a mid-run happy breakdown."""
    rng = np.random.default_rng(21)
    ctx = eu.Context()
    ctx.set_option("patch", 0)             # (grids in their natural ordering: the wave form; the patch form is compared the same way in its own test)
    cases = [(513, 5, 0, False), (1024, 31, 0, False), (4097, 30, 0, False), (10001, 32, 0, False), (3000, 25, 3, False),
             (2500, 30, 0, True), (777, 12, 2, False), (70000, 30, 0, False), (256, 30, 0, False),
             (-40001, 30, 0, False), (-300000, 24, 0, False), (-9000, 31, 4, False), (-25000, 20, 0, True)]
    for n, m, iop, herm in cases:
        grid = n < 0                       # negative size: structured-grid offsets -> wave form of the step
        n = abs(n)
        A = c2_operator(n)
        if grid:
            k = max(9, int(np.sqrt(n)))
            A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
        if herm:
            A = ((A + A.T) * 0.5).tocsr()
        op = eu.MIOperator(A, ctx)
        b = rng.standard_normal(n)
        if n == 777:                      # an invariant subspace of dimension 6: breakdown inside the run
            b = np.zeros(n)
            b[:6] = rng.standard_normal(6)
            A = sp.block_diag([sp.csr_matrix(np.diag(np.arange(1.0, 7.0)) + np.diag(np.ones(5), 1)), c2_operator(n - 6)]).tocsr()
            op = eu.MIOperator(A, ctx)
        out = []
        for overlap in (True, False):
            ctx.set_pipeline_overlap(overlap)
            Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
            eu.arnoldi_(Ks, op, b, m=m, iop=iop, ishermitian=herm)
            w = eu.expv(0.7, op, b, m=m, iop=iop, ishermitian=herm)
            out.append((Ks.m, Ks.wasbreakdown, Ks.H.copy(), Ks.getV().copy(), np.asarray(w).copy()))
        ctx.set_pipeline_overlap(True)
        (m1, bd1, H1, V1, w1), (m2, bd2, H2, V2, w2) = out
        assert (m1, bd1) == (m2, bd2), (n, m, iop, herm)
        assert np.array_equal(H1, H2), (n, m, iop, herm)
        assert np.array_equal(V1[:, : m1 + 1], V2[:, : m2 + 1]), (n, m, iop, herm)
        assert np.array_equal(w1, w2), (n, m, iop, herm)


@pytest.mark.parametrize("case", ["grid2d_small", "grid2d_multi_round", "odd_offsets", "symmetric_grid", "many_diagonals"])
def test_wide_diagonal_operators_wave_form(eu, case, natural_grid_ordering):
    """Operators made of a few diagonals with arbitrary offsets (structured grids) take the wave form of the single-pass
    step: tiles publish their piece of u_j and wait for the tiles their diagonals reach into.  Parity with the oracle
    (small sizes) and with the strict-MGS modular path (every size)."""
    rng = np.random.default_rng(31)
    herm = False
    if case == "grid2d_small":
        n, m, offs = 40_000, 20, [-200, -1, 0, 1, 200]
    elif case == "grid2d_multi_round":          # more tiles than resident workgroups: several rounds of tiles per workgroup
        n, m, offs = 700_000, 30, [-700, -1, 0, 1, 700]
    elif case == "odd_offsets":
        n, m, offs = 123_457, 17, [-3001, -7, 0, 5, 1999]
    elif case == "symmetric_grid":
        n, m, offs, herm = 90_000, 25, [-300, -1, 0, 1, 300], True
    else:
        n, m, offs = 64_000, 12, [-1600, -41, -40, -39, -1, 0, 1, 39, 40, 41, 1600]
    diags = [rng.standard_normal(n - abs(o)) * 0.4 - (2.0 if o == 0 else 0.0) for o in offs]   # variable coefficients
    A = sp.diags(diags, offs, shape=(n, n), format="csr")
    if herm:
        A = ((A + A.T) * 0.5).tocsr()
    b = rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=herm)                      # wave form
    Km = eu.arnoldi(A, b, m=m, ishermitian=herm, ortho="mgs")         # literal MGS, modular path
    assert Ks.m == Km.m and Ks.wasbreakdown == Km.wasbreakdown
    Vm = Km.getV()[:, : m + 1]
    loss = float(np.max(np.abs(Vm.T @ Vm - np.eye(m + 1))))           # rounding differences scale with this
    tol = max(TOL, 10 * loss)
    print("[parity] %-90s loss of orthogonality of the strict-MGS basis: %.3e" % ("wave form: " + case, loss))
    close(Ks.H[: m + 1, :m], Km.H[: m + 1, :m], tol, "wave form %s: H vs strict MGS" % case, mat=True)
    w = eu.expv(0.4, A, b, m=m, ishermitian=herm)
    wm = eu.expv_(np.empty(n), 0.4, Km)
    close(w, wm, tol, "wave form %s: w vs strict MGS" % case)
    close(Ks.getV()[:, : m + 1], Vm, 100 * tol, "wave form %s: V vs strict MGS (max abs)" % case, absolute=True)
    if n <= 100_000:
        Ko = ko.arnoldi(A, b, m=m, ishermitian=herm)
        close(Ks.H[: m + 1, :m], Ko.H[: m + 1, :m], tol, "wave form %s: H vs oracle" % case, mat=True)
        close(w, ko.expv_(np.empty(n), 0.4, Ko), tol, "wave form %s: w vs oracle" % case)


@pytest.mark.parametrize("n,m,band", [(60_000, 20, 1500), (650_000, 30, 2500)])
def test_irregular_banded_operator_wave_form_on_sell_slots(eu, n, m, band):
    """No diagonal structure, but every row's columns lie within `band` of the diagonal (a reordered mesh operator): the
    wave form runs on the SELL slots, a tile waiting for the precomputed range of tiles its columns lie in."""
    rng = np.random.default_rng(41)
    k = 6
    rows = np.repeat(np.arange(n), k)
    cols = rows + rng.integers(-band, band + 1, size=n * k)
    cols = np.clip(cols, 0, n - 1)
    vals = rng.standard_normal(n * k) * 1.0      # (0.15 made the Krylov residual collapse below 1e-9 within 30 steps: H columns of pure rounding noise)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A = (A - 2.0 * sp.eye(n)).tocsr()
    b = rng.standard_normal(n)
    Ks = eu.arnoldi(A, b, m=m, ishermitian=False)
    Km = eu.arnoldi(A, b, m=m, ishermitian=False, ortho="mgs")
    Vm = Km.getV()[:, : m + 1]
    loss = float(np.max(np.abs(Vm.T @ Vm - np.eye(m + 1))))
    print("[parity] %-90s loss of orthogonality of the strict-MGS basis: %.3e" % ("SELL wave form n=%d" % n, loss))
    tol = max(TOL, 10 * loss)
    assert Ks.m == Km.m and Ks.wasbreakdown == Km.wasbreakdown
    close(Ks.H[: m + 1, :m], Km.H[: m + 1, :m], tol, "SELL wave form n=%d: H vs strict MGS" % n, mat=True)
    w = eu.expv(0.4, A, b, m=m, ishermitian=False)
    close(w, eu.expv_(np.empty(n), 0.4, Km), TOL, "SELL wave form n=%d: w vs strict MGS" % n)
    if n <= 100_000:
        close(w, ko.expv(0.4, A, b, m=m, ishermitian=False), TOL, "SELL wave form n=%d: w vs oracle" % n)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("kind", ["dense", "csr"])
def test_exact_breakdown_leaves_no_nan_for_the_next_factorisation(eu, T, kind):
    """An EXACT happy breakdown (beta = 0) makes v_{m+1} = y / beta NaN, in the reference too (arnoldi.jl:306, :399: the division
    comes before the test).  The padding rows of that column must stay zero (0 / 0 is NaN as well) and the stale NaN column must
    not reach the next factorisation on the same subspace.  Found by tests/fuzz_parity.py (seed 11, case 2496: n = 1, Float32)."""
    A = np.array([[-0.55]], dtype=T)
    Ah = np.array([[-0.7]], dtype=T)
    b = np.array([0.36], dtype=T)
    if kind == "csr":
        A, Ah = sp.csr_matrix(A), sp.csr_matrix(Ah)
    for ortho in ("mgs", "lowsync"):
        Ks = eu.KrylovSubspace(T, None, 1, 31)
        eu.arnoldi_(Ks, Ah, b, m=31, ishermitian=True, ortho=ortho)
        assert Ks.m == 1 and Ks.wasbreakdown
        eu.arnoldi_(Ks, A, b, m=31, iop=7, ishermitian=False, ortho=ortho)
        H = np.asarray(Ks.getH())
        assert Ks.m <= 2 and Ks.wasbreakdown and np.isfinite(H[:Ks.m, :Ks.m]).all(), (Ks.m, H[:3, :3])
        w = eu.expv(0.7, Ah, b, m=31, ishermitian=True, ortho=ortho)
        w2 = eu.expv(0.7, A, b, m=31, ishermitian=False, ortho=ortho)
        tol = 1e-6 if np.dtype(T).itemsize <= 8 and np.dtype(T) != np.float64 else 1e-14
        close(np.asarray(w).astype(np.complex128), np.exp(0.7 * -0.7) * 0.36, tol, "n=1 %s %s %s after an exact breakdown: expv (Lanczos)" % (np.dtype(T).name, kind, ortho))
        close(np.asarray(w2).astype(np.complex128), np.exp(0.7 * -0.55) * 0.36, tol, "n=1 %s %s %s after an exact breakdown: expv (Arnoldi)" % (np.dtype(T).name, kind, ortho))


@pytest.mark.gpu
def test_randomised_parity_hunt_short():
    """Six seconds of tests/fuzz_parity.py with a fixed seed (~350 random cases over element types, sizes, operator structures, calls
    and options against the oracle): no failure, no exception.  profiles/r03_fuzz_parity.txt has the long runs and what they found."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_LARGE="0.02")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "6", "20260928"], cwd=root, capture_output=True, text=True,
                       timeout=600, env=env)
    last = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(last)
    assert p.returncode == 0 and r["failures"] == 0 and r["cases"] > 100, p.stdout[-3000:] + p.stderr[-2000:]


# ------------------------------------------------------------------ reordered operators (reorder.h) --------
def _shuffle(A, seed):
    q = np.random.default_rng(seed).permutation(A.shape[0])
    return A[q][:, q].tocsr()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["band_f64", "band_c64", "band_f32", "grid_f64", "band_csc"])
def test_reordered_operator_matches_the_oracle_on_the_natural_ordering(eu, case):
    """an unstructured operator is stored as P A P' (reverse Cuthill-McKee at creation) when that puts it on
    the single-pass step; vectors are permuted on entry and exit, the basis stays permuted.  H and beta do not depend on the
    ordering, V and every result come back in the caller's ordering: all of it against the oracle run on the caller's matrix, at
    the fixed 1e-12 bars (fp32: the fp32 bars).  arnoldi! + getH / getV, expv!, phiv!, the whole-call expv (host and device
    vectors), mul!, a continuation after the basis has been looked at, lanczos!, the error-estimate mode, phiv_timestep! and
    kiops."""
    import torch
    rng = np.random.default_rng(21)
    T = {"band_f64": np.float64, "band_c64": np.complex128, "band_f32": np.float32, "grid_f64": np.float64, "band_csc": np.float64}[case]
    cplx = np.dtype(T).kind == "c"
    if case == "grid_f64":
        k = 600
        n = k * k                                                  # 360 000 rows: > 400 tiles, so the reach decides the form
        A0 = sp.diags([0.7, 1.1, -4.0, 0.9, 1.3], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
        m = 12
    else:
        n = 70_001
        A0 = c2_operator(n)
        m = 20                                                     # (round 5: complex windows beyond 15 columns stay on the single-pass step)
    A = _shuffle(A0 * ((1 + 0.25j) if cplx else 1.0), 8).astype(T)
    if case == "band_csc":
        A = A.tocsc()
    A64 = A.astype(np.complex128 if cplx else np.float64)
    b = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
    b64 = b.astype(A64.dtype)
    tol = 2e-5 if np.dtype(T).itemsize <= 4 else TOL
    ctx = eu.Context()
    ctx.set_option("patch", 0)                 # (reverse Cuthill-McKee; with patch = 1 the shuffled grid is cut into patches: test_patch_form_…)
    op = eu.MIOperator(A, ctx)
    ri = op.reorder_info
    assert ri["reordered"] and ri["bandwidth_after"] < ri["bandwidth_before"] // 50, ri
    plain_ctx = eu.Context()
    plain_ctx.set_option("reorder", 0)
    assert not eu.MIOperator(A, plain_ctx).reorder_info["reordered"]
    # mul!
    close(np.asarray(op.matvec(b)), A64 @ b64, 5e-6 if tol > TOL else 1e-14, "reordered %s: mul! vs scipy" % case)
    # arnoldi! -> H, beta, V
    Ks = eu.KrylovSubspace(T, T, n, m + 6, 0, ctx)
    eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
    Ko = ko.KrylovSubspace(A64.dtype.type, A64.dtype.type, n, m + 6)
    ko.arnoldi_(Ko, A64.tocsr(), b64, m=m, ishermitian=False)
    assert Ks.m == Ko.m == m
    assert abs(Ks.beta - Ko.beta) <= (1e-6 if tol > TOL else 1e-14) * Ko.beta
    close(np.asarray(Ks.getH()).astype(A64.dtype), Ko.getH(), tol, "reordered %s: H of arnoldi! vs oracle (natural ordering)" % case, mat=True)
    # expv! / phiv! BEFORE anybody looked at V: the basis is still in the stored ordering
    w = eu.expv_(np.empty(n, dtype=T), 0.7, Ks)
    close(np.asarray(w).astype(A64.dtype), ko.expv_(np.empty(n, dtype=A64.dtype), 0.7, Ko), tol, "reordered %s: expv! vs oracle" % case)
    W = eu.phiv_(np.empty((n, 3), dtype=T, order="F"), 0.7, Ks, 2)
    close(np.asarray(W).astype(A64.dtype), ko.phiv_(np.empty((n, 3), dtype=A64.dtype), 0.7, Ko, 2), 10 * tol, "reordered %s: phiv! k=2 vs oracle" % case)
    wd = torch.empty(n, dtype=torch.as_tensor(b).dtype, device="cuda")
    eu.expv_(wd, 0.7, Ks)
    ctx.sync()
    close(wd.cpu().numpy().astype(A64.dtype), np.asarray(w).astype(A64.dtype), 0.0 if tol == TOL else 1e-7, "reordered %s: expv! into a device vector == into a host vector" % case)
    # getV: rows back in the caller's ordering (the basis is converted in place) ...
    close(np.asarray(Ks.getV()).astype(A64.dtype), Ko.getV(), tol, "reordered %s: V vs oracle (max abs)" % case, absolute=True)
    # ... and everything still works afterwards: expv! from the converted basis, a continuation (converted back)
    w2 = eu.expv_(np.empty(n, dtype=T), 0.7, Ks)
    close(np.asarray(w2).astype(A64.dtype), np.asarray(w).astype(A64.dtype), 1e-6 if tol > TOL else 1e-14, "reordered %s: expv! after getV" % case)
    if case in ("band_f64", "grid_f64"):
        eu.arnoldi_(Ks, op, b, m=m + 6, init=m, ishermitian=False)
        ko.arnoldi_(Ko, A64.tocsr(), b64, m=m + 6, init=m, ishermitian=False)
        close(Ks.getH(), Ko.getH(), TOL, "reordered %s: H after a continuation (init = m) vs oracle" % case, mat=True)
        close(Ks.getV(), Ko.getV(), TOL, "reordered %s: V after a continuation vs oracle (max abs)" % case, absolute=True)
    # whole-call expv: host vectors and device vectors
    wv = eu.expv(0.7, op, b, m=m, ishermitian=False)
    wo = ko.expv(0.7, A64.tocsr(), b64, m=m, ishermitian=False)
    close(np.asarray(wv).astype(A64.dtype), wo, tol, "reordered %s: expv(t, A, b) vs oracle" % case)
    assert "pipeline" in " ".join(eu.expv.last_stats["path"]) or "single" in " ".join(eu.expv.last_stats["path"]), eu.expv.last_stats
    bd = torch.as_tensor(b, device="cuda")
    out = torch.empty_like(bd)
    eu.expv(0.7, op, bd, m=m, ishermitian=False, out=out)
    ctx.sync()
    close(out.cpu().numpy().astype(A64.dtype), wo, tol, "reordered %s: expv with device vectors vs oracle" % case)


@pytest.mark.gpu
def test_reordered_operator_drivers_and_value_updates(eu):
    """The drivers on a reordered operator -- lanczos! / error-estimate mode on a symmetric one, adaptive phiv_timestep!, kiops --
    against the oracle on the caller's ordering, and a values-only update through the caller's (CSC) entry order."""
    rng = np.random.default_rng(5)
    n = 50_000
    As = _shuffle(c2_operator(n, sym=True), 3)
    b = rng.standard_normal(n)
    ctx = eu.Context()
    ops = eu.MIOperator(As.tocsc(), ctx)
    assert ops.reorder_info["reordered"] and ops.ishermitian
    w = eu.expv(0.5, ops, b, m=25)
    close(w, ko.expv(0.5, As, b, m=25), TOL, "reordered symmetric operator: expv (Lanczos) vs oracle")
    we = eu.expv(0.5, ops, b, m=30, mode="error_estimate", rtol=1e-9)
    woe = ko.expv(0.5, As, b, m=30, mode="error_estimate", rtol=1e-9)
    close(we, woe, 1e-11, "reordered symmetric operator: error-estimate mode vs oracle")
    A = _shuffle(c2_operator(n), 4)
    op = eu.MIOperator(A.tocsc(), ctx)
    B = np.asfortranarray(rng.standard_normal((n, 3)))
    st, so = {}, {}
    ts = np.array([0.4, 1.0])
    U = eu.phiv_timestep(ts.copy(), op, B, adaptive=True, tol=1e-8, stats=st)
    Uo = ko.phiv_timestep(ts.copy(), A, B, adaptive=True, tol=1e-8, stats=so)
    assert (st["num_timesteps"], st["matvecs"], st["m"]) == (so["num_timesteps"], so["matvecs"], so["m"]), (st, so)
    close(U, Uo, 1e-11, "reordered operator: adaptive phiv_timestep (K = 2, two snapshots) vs oracle")
    wk, sk = eu.kiops(1.0, op, B, ishermitian=False)
    wko, sko = ko.kiops(1.0, A, B, ishermitian=False)
    assert tuple(sk) == tuple(sko), (sk, sko)
    close(wk, wko, 1e-10, "reordered operator: kiops with three columns vs oracle")
    # values-only update in the caller's entry order (CSC here): the map goes through CSC -> CSR -> P A P'
    Ac = A.tocsc()
    Ac.sort_indices()
    op2 = eu.MIOperator(Ac, ctx)
    Ac2 = Ac.copy()
    Ac2.data = Ac.data * (1.0 + 0.3 * rng.random(Ac.nnz))
    op2.update_values(Ac2)
    assert op2.reorder_info["reordered"]
    close(eu.expv(0.6, op2, b, m=20, ishermitian=False), ko.expv(0.6, Ac2.tocsr(), b, m=20, ishermitian=False), TOL,
          "reordered operator after update_values vs oracle on the new matrix")
    close(op2.opnorm_inf, float(np.max(np.abs(Ac2).sum(axis=1))), 1e-14, "reordered operator: opnorm(A, Inf) after update_values")


def _grid_operator(case, rng):
    """2-D grid stencils for the patch form: (A, m).  pure: five full diagonals (the bench's operator: the +-1 diagonals run across
    the row ends); laplace: a 5-point operator with per-entry coefficients and NO entries across row ends (kron structure); nine: a
    9-point stencil (offsets +-k, +-k+-1); ragged: the last grid row is incomplete; big: n = 10^6, several tiles per workgroup."""
    def five_point(k, rows, wrap):
        n = k * rows
        i = np.arange(n)
        parts = []
        for off in (-k, -1, 0, 1, k):
            j = i + off
            ok = (j >= 0) & (j < n)
            if not wrap and abs(off) == 1:
                ok &= (j // k) == (i // k)
            v = (-4.0 if off == 0 else 1.0) + 0.3 * rng.standard_normal(n)
            parts.append(sp.csr_matrix((v[ok], (i[ok], j[ok])), shape=(n, n)))
        return sum(parts).tocsr()
    if case in ("pure_f64", "pure_f32", "pure_serial"):
        k, rows = 320, 300
        n = k * rows
        return sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr"), 30
    if case == "laplace_f64":
        return five_point(250, 260, False), 20
    if case == "nine_f64":
        k, rows = 200, 333
        n = k * rows
        offs = [-k - 1, -k, -k + 1, -1, 0, 1, k - 1, k, k + 1]
        return sp.diags([0.05, 0.4, -0.07, 1.1, -3.0, 0.9, 0.06, 0.5, 0.03], offs, shape=(n, n), format="csr"), 16
    if case == "ragged_f64":
        k = 131
        n = k * 257 + 77
        return sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr"), 24
    if case == "big_f64":
        k = 1000
        return sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(k * k, k * k), format="csr"), 12
    if case in ("mesh_f64", "mesh_f32", "trimesh_components_f64"):
        # meshes in a RANDOM numbering (creation cuts them into patches from breadth-first distances: reorder.h mesh_patches): a planar
        # 5-point mesh; two triangulated pieces of different size + isolated unknowns
        def planar(k, rows, tri):
            n = k * rows
            i = np.arange(n)
            parts = []
            for dr, dc in [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)] + ([(1, 1), (-1, -1)] if tri else []):
                r, c = i // k + dr, i % k + dc
                ok = (r >= 0) & (r < rows) & (c >= 0) & (c < k)
                v = (-3.0 if (dr, dc) == (0, 0) else 0.6) + 0.3 * rng.standard_normal(n)
                parts.append(sp.csr_matrix((v[ok], (i[ok], (r * k + c)[ok])), shape=(n, n)))
            return sum(parts).tocsr()
        if case == "trimesh_components_f64":
            A = sp.block_diag([planar(210, 190, True), planar(97, 160, True), -0.7 * sp.identity(300, format="csr")], format="csr")
        else:
            A = planar(330, 290, False)
        q = rng.permutation(A.shape[0])
        A = A[q][:, q].tocsr()
        A.sort_indices()
        return A, 24
    raise ValueError(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pure_f64", "pure_serial", "pure_f32", "laplace_f64", "nine_f64", "ragged_f64", "big_f64", "mesh_f64", "mesh_f32",
                                  "trimesh_components_f64"])
def test_patch_form_of_the_single_pass_step(eu, case):
    """a 2-D grid stencil stored in a grid-patch ordering (context option patch = 1: a tile of the single-pass step
    is a 16 x 32 patch of the grid, the ring of rows around it is recomputed like the banded form's halo -- no per-tile flags).  The
    caller sees the natural ordering: H, beta, V, expv!, phiv!, a continuation, the whole-call expv with host and device vectors,
    mul!, a short window (iop = 3) and a values-only update, all against the oracle on the caller's matrix at the fixed bars."""
    import torch
    rng = np.random.default_rng(33)
    T = np.float32 if case in ("pure_f32", "mesh_f32") else np.float64
    A0, m = _grid_operator(case, rng)
    A = A0.astype(T)
    n = A.shape[0]
    A64 = A.astype(np.float64)
    b = rng.standard_normal(n).astype(T)
    b64 = b.astype(np.float64)
    tol = 2e-5 if T == np.float32 else TOL
    ctx = eu.Context()
    ctx.set_option("patch", 1)
    if case == "pure_serial":
        ctx.set_option("pipeline_serial", 1)
    op = eu.MIOperator(A, ctx)
    pi = op.patch_info
    assert pi["patch_form"] and op.reorder_info["reordered"] and pi["longest_ring"] <= 256, pi
    if case == "big_f64":
        assert pi["tiles"] == 1954 and pi["tiles_ring_over_128"] == 0 and 90 < pi["mean_ring"] < 100, pi
    close(np.asarray(op.matvec(b)), A64 @ b64, 5e-6 if tol > TOL else 1e-14, "patch form %s: mul! vs scipy" % case)
    Ks = eu.KrylovSubspace(T, T, n, m + 6, 0, ctx)
    eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
    Ko = ko.KrylovSubspace(np.float64, np.float64, n, m + 6)
    ko.arnoldi_(Ko, A64, b64, m=m, ishermitian=False)
    assert Ks.m == Ko.m == m
    close(np.asarray(Ks.getH()).astype(np.float64), Ko.getH(), tol, "patch form %s: H of arnoldi! vs oracle" % case, mat=True)
    w = eu.expv_(np.empty(n, dtype=T), 0.7, Ks)
    close(np.asarray(w).astype(np.float64), ko.expv_(np.empty(n), 0.7, Ko), tol, "patch form %s: expv! vs oracle" % case)
    W = eu.phiv_(np.empty((n, 3), dtype=T, order="F"), 0.7, Ks, 2)
    close(np.asarray(W).astype(np.float64), ko.phiv_(np.empty((n, 3)), 0.7, Ko, 2), 10 * tol, "patch form %s: phiv! k=2 vs oracle" % case)
    close(np.asarray(Ks.getV()).astype(np.float64), Ko.getV(), tol, "patch form %s: V vs oracle (max abs)" % case, absolute=True)
    if T == np.float64 and case != "big_f64":
        eu.arnoldi_(Ks, op, b, m=m + 6, init=m, ishermitian=False)
        ko.arnoldi_(Ko, A64, b64, m=m + 6, init=m, ishermitian=False)
        close(Ks.getH(), Ko.getH(), TOL, "patch form %s: H after a continuation (init = m) vs oracle" % case, mat=True)
        close(Ks.getV(), Ko.getV(), TOL, "patch form %s: V after a continuation vs oracle (max abs)" % case, absolute=True)
    wo = ko.expv(0.7, A64, b64, m=m, ishermitian=False)
    close(np.asarray(eu.expv(0.7, op, b, m=m, ishermitian=False)).astype(np.float64), wo, tol, "patch form %s: expv(t, A, b) vs oracle" % case)
    assert "patch" in eu.expv.last_stats["path"] and ("overlapped" in eu.expv.last_stats["path"]) == (case != "pure_serial"), eu.expv.last_stats
    bd = torch.as_tensor(b, device="cuda")
    out = torch.empty_like(bd)
    eu.expv(0.7, op, bd, m=m, ishermitian=False, out=out)
    ctx.sync()
    close(out.cpu().numpy().astype(np.float64), wo, tol, "patch form %s: expv with device vectors vs oracle" % case)
    if case != "pure_serial":              # one launch after the other: same kernels, same sums -- bit for bit
        ctx.set_pipeline_overlap(False)
        out2 = torch.empty_like(bd)
        eu.expv(0.7, op, bd, m=m, ishermitian=False, out=out2)
        ctx.sync()
        ctx.set_pipeline_overlap(True)
        assert "overlapped" not in eu.expv.last_stats["path"] and torch.equal(out, out2), "patch form %s: overlapped and serial runs differ" % case
    if case == "big_f64":
        return
    # a short orthogonalisation window (incomplete orthogonalisation, iop = 3)
    close(np.asarray(eu.expv(0.5, op, b, m=m, iop=3, ishermitian=False)).astype(np.float64), ko.expv(0.5, A64, b64, m=m, iop=3, ishermitian=False),
          10 * tol, "patch form %s: expv with iop = 3 vs oracle" % case)
    assert "patch" in eu.expv.last_stats["path"], eu.expv.last_stats
    # new values on the same pattern, in the caller's entry order
    A2 = A.copy()
    A2.data = (A.data * (1.0 + 0.2 * rng.random(A.nnz))).astype(T)
    op.update_values(A2)
    close(np.asarray(eu.expv(0.6, op, b, m=m, ishermitian=False)).astype(np.float64), ko.expv(0.6, A2.astype(np.float64), b64, m=m, ishermitian=False), tol,
          "patch form %s: expv after update_values vs oracle on the new matrix" % case)
    assert "patch" in eu.expv.last_stats["path"], eu.expv.last_stats


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_banded_operator_without_a_diagonal_form_runs_on_tile_local_columns(eu, T):
    """A banded operator with more than 8 distinct offsets has no diagonal (DIA) form; its halo form on SELL slots reads 4 bytes of
    column index per entry from HBM.  With option patch (default) the same operator -- in its own ordering, nothing is permuted --
    runs on the patch form with the halo as its ring and tile-local column indices whose equal blocks are stored once (0.649 ->
    0.677 of the contract on the C2 pattern).  Both against the oracle; kiops (augmented operator) and a continuation too."""
    rng = np.random.default_rng(41)
    n, m = 150_001, 26
    offs = [-8, -6, -5, -3, -1, 0, 1, 2, 4, 7]
    d = [(0.1 + 0.05 * rng.random(n - abs(o))) * (1 if o else -6.0) for o in offs]
    A = sp.diags(d, offs, shape=(n, n), format="csr").astype(T)
    A64 = A.astype(np.float64)
    b = rng.standard_normal(n).astype(T)
    b64 = b.astype(np.float64)
    tol = 2e-5 if T == np.float32 else TOL
    wo = ko.expv(0.7, A64, b64, m=m, ishermitian=False)
    res = {}
    for patch in (1, 0):
        ctx = eu.Context()
        ctx.set_option("patch", patch)
        op = eu.MIOperator(A, ctx)
        assert not op.reorder_info["reordered"] and op.patch_info["patch_form"] == bool(patch), (op.reorder_info, op.patch_info)
        w = np.asarray(eu.expv(0.7, op, b, m=m, ishermitian=False)).astype(np.float64)
        path = eu.expv.last_stats["path"]
        assert "pipeline" in path and ("patch" in path) == bool(patch), path
        close(w, wo, tol, "banded operator, ten offsets, patch = %d (%s): expv vs oracle" % (patch, np.dtype(T).name))
        res[patch] = w
        if patch and T == np.float64:
            pi = op.patch_info
            assert pi["longest_ring"] <= 16 and pi["column_indices_stored"] < A.nnz // 50, pi
            Ks = eu.KrylovSubspace(T, T, n, m + 4, 0, ctx)
            eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
            eu.arnoldi_(Ks, op, b, m=m + 4, init=m, ishermitian=False)
            Ko = ko.KrylovSubspace(np.float64, np.float64, n, m + 4)
            ko.arnoldi_(Ko, A64, b64, m=m, ishermitian=False)
            ko.arnoldi_(Ko, A64, b64, m=m + 4, init=m, ishermitian=False)
            close(Ks.getH(), Ko.getH(), TOL, "banded operator on tile-local columns: H after a continuation vs oracle", mat=True)
            wk, sk = eu.kiops(0.8, op, b, ishermitian=False)
            wko, sko = ko.kiops(0.8, A64, b64, ishermitian=False)
            assert tuple(sk) == tuple(sko), (sk, sko)
            close(wk, wko, 1e-10, "banded operator on tile-local columns: kiops vs oracle")
    close(res[1], res[0], 10 * tol, "patch form == halo form on SELL slots")


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", ["ten_offsets", "band40", "grid"])
def test_rows_stored_out_of_order_or_with_repeated_entries_through_the_c_abi(eu, pattern):
    """expv_mi_op_create_csr takes rows as they are: columns in any order, an entry given twice (its values add up, as in
    SparseArrays' mul!).  The Python front end always sorts, so this goes through the C ABI: the same matrix with every row's entries
    reversed and one entry per row split in two must give the results of the tidy one -- whichever storage form and ordering creation
    picks for it (tile-local columns are only built for rows with strictly ascending columns; reordered operators are rewritten)."""
    import ctypes as C
    from exponentialutilities_jl_amd import _lib as L
    rng = np.random.default_rng(53)
    k = 96
    n = k * 700
    offs = {"ten_offsets": [-8, -6, -5, -3, -1, 0, 1, 2, 4, 7], "band40": [-40, -1, 0, 1, 40], "grid": [-k, -1, 0, 1, k]}[pattern]
    A = sp.diags([(0.2 + 0.1 * rng.random(n - abs(o))) * (-2.0 if o == 0 else 1.0) for o in offs], offs, shape=(n, n), format="csr")
    A.sort_indices()
    b = rng.standard_normal(n)
    ctx = eu.Context()
    want = np.asarray(eu.expv(0.5, eu.MIOperator(A, ctx), b, m=18, ishermitian=False))
    close(want, ko.expv(0.5, A, b, m=18, ishermitian=False), TOL, "tidy rows (%s): expv vs oracle" % pattern)
    # every row reversed; the row's first entry split into two halves at both ends of the row
    ip, ix, vv = [0], [], []
    for r in range(n):
        c = A.indices[A.indptr[r]:A.indptr[r + 1]][::-1]
        v = A.data[A.indptr[r]:A.indptr[r + 1]][::-1]
        ix.extend([c[0]] + list(c[1:]) + [c[0]])
        vv.extend([0.25 * v[0]] + list(v[1:]) + [0.75 * v[0]])
        ip.append(len(ix))
    ip, ix, vv = np.asarray(ip, dtype=np.int32), np.asarray(ix, dtype=np.int32), np.asarray(vv, dtype=np.float64)
    lib = L.load()
    h = C.c_void_p()
    assert lib.expv_mi_op_create_csr(ctx._h, L.F64, n, ip.ctypes.data, ix.ctypes.data, vv.ctypes.data, 4, 0, C.byref(h)) == 0
    try:
        w = np.empty(n)
        o = L.ArnoldiOpts()
        lib.expv_mi_arnoldi_opts_default(C.byref(o))
        o.m, o.ishermitian = 18, 0
        st = L.ExpvStats()
        assert lib.expv_mi_expv(ctx._h, h, 0.5, 0.0, b.ctypes.data, L.HOST, w.ctypes.data, L.HOST, L.F64, C.byref(o), C.byref(st)) == 0
        close(w, want, 1e-13, "rows reversed + a repeated entry (%s): expv == the tidy matrix" % pattern)
        y = np.empty(n)
        assert lib.expv_mi_op_apply(h, b.ctypes.data, L.HOST, y.ctypes.data, L.HOST) == 0
        close(y, A @ b, 1e-14, "rows reversed + a repeated entry (%s): mul!" % pattern)
    finally:
        lib.expv_mi_op_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.float32, np.complex128])
def test_wide_band_operator_in_its_own_ordering_on_the_patch_form(eu, T):
    """A band wider than the halo form's 8 rows but within 64 -- a thin 2-D grid with rows of 40 cells, coefficients varying -- ran the
    wave form up to round 3 (the complex types: the two-kernel step).  The patch form takes it in its own ordering (nothing
    permuted): the ring of a tile is the 2 x 40 rows above and below it.  H, expv and a short window against the oracle."""
    rng = np.random.default_rng(47)
    cplx = np.dtype(T).kind == "c"
    k = 30 if cplx else 40                 # (the band may be an eighth of a tile: 64 rows for Float64, 32 for ComplexF64)
    n, m = k * 2600 + 17, 14
    d = [(0.3 + 0.1 * rng.random(n - abs(o))) * (-2.0 if o == 0 else 1.0) * ((1 + 0.2j) if cplx else 1.0) for o in (-k, -1, 0, 1, k)]
    A = sp.diags(d, [-k, -1, 0, 1, k], shape=(n, n), format="csr").astype(T)
    T64 = np.complex128 if cplx else np.float64
    A64 = A.astype(T64)
    b = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
    b64 = b.astype(T64)
    tol = 2e-5 if T == np.float32 else TOL
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    pi = op.patch_info
    assert pi["patch_form"] and not op.reorder_info["reordered"] and 2 * k <= pi["longest_ring"] <= 2 * k + 4, (pi, op.reorder_info)
    w = np.asarray(eu.expv(0.6, op, b, m=m, ishermitian=False)).astype(T64)
    assert "patch" in eu.expv.last_stats["path"], eu.expv.last_stats
    close(w, ko.expv(0.6, A64, b64, m=m, ishermitian=False), tol, "band of 40 rows (%s), own ordering, patch form: expv vs oracle" % np.dtype(T).name)
    Ks = eu.arnoldi(op, b, m=m, ishermitian=False)
    close(np.asarray(Ks.getH()).astype(T64), ko.arnoldi(A64, b64, m=m, ishermitian=False).getH(), tol, "band of 40 rows (%s): H vs oracle" % np.dtype(T).name, mat=True)
    close(np.asarray(eu.expv(0.6, op, b, m=m, iop=2, ishermitian=False)).astype(T64), ko.expv(0.6, A64, b64, m=m, iop=2, ishermitian=False), 10 * tol,
          "band of 40 rows (%s): expv with iop = 2 vs oracle" % np.dtype(T).name)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["schroedinger_c128", "complex_grid_c128", "complex_grid_c64", "complex_banded_c128", "complex_mesh_c128"])
def test_patch_form_complex_element_types(eu, case):
    """The patch form for the complex element types (tiles of 256 ComplexF64 / 512 ComplexF32 rows = 16 x 16 / 16 x 32 patches): the
    only single-pass form they have beyond 8 full diagonals.  A Schroedinger-type problem -- a REAL symmetric 2-D grid operator, a
    complex vector and an imaginary time (T = promote(eltype A, eltype b), Lanczos with complex t: krylov_phiv.jl:252-280) --, a
    complex non-Hermitian grid stencil (Arnoldi, windows up to 15), its ComplexF32 version, a complex banded operator with ten offsets
    (own ordering) and a complex mesh numbered at random; H, expv!, the whole call and the error-estimate mode against the oracle."""
    rng = np.random.default_rng(43)
    T = np.complex64 if case == "complex_grid_c64" else np.complex128
    tol = 2e-5 if T == np.complex64 else TOL
    k, rows = 180, 150
    n = k * rows
    herm = False
    t = 0.7
    if case == "schroedinger_c128":
        pot = 0.3 * rng.random(n)
        A0 = sp.diags([np.full(n - k, 1.0), np.full(n - 1, 1.0), -4.0 + pot, np.full(n - 1, 1.0), np.full(n - k, 1.0)], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
        herm, t, m = True, -0.6j, 28
    elif case in ("complex_grid_c128", "complex_grid_c64"):
        A0 = (sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr") * (1 + 0.25j)).tocsr()
        m = 14
    elif case == "complex_banded_c128":
        offs = [-8, -6, -5, -3, -1, 0, 1, 2, 4, 7]
        A0 = sp.diags([(0.1 + 0.05 * rng.random(n - abs(o))) * (1 if o else -6.0) * (1 + 0.3j) for o in offs], offs, shape=(n, n), format="csr")
        m = 12
    else:
        i = np.arange(n)
        parts = []
        for dr, dc in [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)]:
            r, c = i // k + dr, i % k + dc
            ok = (r >= 0) & (r < rows) & (c >= 0) & (c < k)
            v = ((-3.0 if (dr, dc) == (0, 0) else 0.6) + 0.2 * rng.standard_normal(n)) * (1 - 0.2j)
            parts.append(sp.csr_matrix((v[ok], (i[ok], (r * k + c)[ok])), shape=(n, n)))
        q = rng.permutation(n)
        A0 = sum(parts).tocsr()[q][:, q].tocsr()
        A0.sort_indices()
        m = 13
    A = A0.astype(T if case != "schroedinger_c128" else np.float64)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(T)
    A128, b128 = A.astype(np.complex128), b.astype(np.complex128)
    ctx = eu.Context()
    op = eu.MIOperator(A.astype(T), ctx)          # (the Schroedinger case: the operator in the vectors' element type, as the front end would)
    assert op.patch_info["patch_form"], op.patch_info
    assert op.reorder_info["reordered"] == (case != "complex_banded_c128"), op.reorder_info
    w = np.asarray(eu.expv(t, op, b, m=m, ishermitian=herm)).astype(np.complex128)
    path = eu.expv.last_stats["path"]
    assert "patch" in path and "pipeline" in path, path
    close(w, ko.expv(t, A128, b128, m=m, ishermitian=herm), tol, "patch form, %s: expv vs oracle" % case)
    Ks = eu.KrylovSubspace(T, np.float32 if (herm and T == np.complex64) else (np.float64 if herm else T), n, m, 0, ctx)
    eu.arnoldi_(Ks, op, b, m=m, ishermitian=herm)
    Ko = ko.arnoldi(A128, b128, m=m, ishermitian=herm)
    close(np.asarray(Ks.getH()).astype(np.complex128), Ko.getH(), tol, "patch form, %s: H vs oracle" % case, mat=True)
    close(np.asarray(eu.expv_(np.empty(n, dtype=T), t, Ks)).astype(np.complex128), ko.expv_(np.empty(n, dtype=np.complex128), t, Ko), tol,
          "patch form, %s: expv! vs oracle" % case)
    ctx2 = eu.Context()
    ctx2.set_option("patch", 0)
    w0 = np.asarray(eu.expv(t, eu.MIOperator(A.astype(T), ctx2), b, m=m, ishermitian=herm)).astype(np.complex128)
    assert "patch" not in eu.expv.last_stats["path"]
    close(w, w0, 10 * tol, "patch form, %s: == the natural-ordering path" % case)
    if case == "complex_grid_c128":      # kiops with complex operands (this build's extension): the augmented operator on the patch form
        wk, sk = eu.kiops(0.8, op, b, allow_complex=True, ishermitian=False)
        wko, sko = ko.kiops(0.8, A128, b128, allow_complex=True, ishermitian=False)
        assert tuple(sk) == tuple(sko), (sk, sko)
        close(wk, wko, 1e-10, "patch form, complex grid: kiops vs oracle")
    if case == "schroedinger_c128":
        we = eu.expv(t, op, b, m=30, mode="error_estimate", rtol=1e-9)
        close(we, ko.expv(t, A128, b128, m=30, mode="error_estimate", rtol=1e-9), 1e-11, "patch form, Schroedinger: error-estimate mode vs oracle")


@pytest.mark.gpu
def test_patch_form_drivers(eu):
    """The drivers on an operator stored in the grid-patch ordering: lanczos! and the error-estimate mode on a symmetric stencil
    (window 2 on the patch form), adaptive phiv_timestep! and kiops (augmented operator: the two-kernel step on the stored ordering),
    each against the oracle on the caller's ordering."""
    rng = np.random.default_rng(34)
    k, rows = 256, 200
    n = k * rows
    As = sp.diags([0.5, 1.0, -3.0, 1.0, 0.5], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    b = rng.standard_normal(n)
    ctx = eu.Context()
    ctx.set_option("patch", 1)
    ops = eu.MIOperator(As, ctx)
    assert ops.patch_info["patch_form"] and ops.ishermitian
    close(eu.expv(0.5, ops, b, m=25), ko.expv(0.5, As, b, m=25), TOL, "patch form, symmetric stencil: expv (Lanczos) vs oracle")
    assert "patch" in eu.expv.last_stats["path"], eu.expv.last_stats
    close(eu.expv(0.5, ops, b, m=30, mode="error_estimate", rtol=1e-9), ko.expv(0.5, As, b, m=30, mode="error_estimate", rtol=1e-9), 1e-11,
          "patch form, symmetric stencil: error-estimate mode vs oracle")
    A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    op = eu.MIOperator(A.tocsc(), ctx)
    assert op.patch_info["patch_form"]
    B = np.asfortranarray(rng.standard_normal((n, 3)))
    st, so = {}, {}
    ts = np.array([0.4, 1.0])
    U = eu.phiv_timestep(ts.copy(), op, B, adaptive=True, tol=1e-8, stats=st)
    Uo = ko.phiv_timestep(ts.copy(), A, B, adaptive=True, tol=1e-8, stats=so)
    assert (st["num_timesteps"], st["matvecs"], st["m"]) == (so["num_timesteps"], so["matvecs"], so["m"]), (st, so)
    close(U, Uo, 1e-11, "patch form: adaptive phiv_timestep (K = 2, two snapshots) vs oracle")
    wk, sk = eu.kiops(1.0, op, B, ishermitian=False)
    wko, sko = ko.kiops(1.0, A, B, ishermitian=False)
    assert tuple(sk) == tuple(sko), (sk, sko)
    close(wk, wko, 1e-10, "patch form: kiops with three columns vs oracle")
    wk1, sk1 = eu.kiops(0.8, op, b, ishermitian=False)            # p = 1: the augmented operator on the patch form of the step
    wko1, sko1 = ko.kiops(0.8, A, b, ishermitian=False)
    assert tuple(sk1) == tuple(sko1), (sk1, sko1)
    close(wk1, wko1, 1e-10, "patch form: kiops with a vector vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["real", "complex", "float32", "real_csc_update"])
def test_irregular_rows_column_blocked_form(eu, case):
    """Round 4: irregular rows of an operator whose vector does not fit an XCD's L2 next to the streams (n * sizeof(T) >= 2 MB) are
    stored in the column-blocked form -- every entry packed by (2 MB column block, row), chunks of 256 entries per wave, per-row sums
    through LDS, one partial vector per block, no SELL slots (kernels.hip: k_spmv_cbf).  mul!, H of arnoldi!, expv against the
    oracle at the fixed bars; a row of several thousand entries (several chunks in one block: the partial-sum path), an empty row,
    complex and Float32 values, CSC input and a values-only update; the small-operator path (SELL cut + overflow pass) keeps its own
    tests above."""
    T = {"real": np.float64, "complex": np.complex128, "float32": np.float32, "real_csc_update": np.float64}[case]
    cplx = np.dtype(T).kind == "c"
    n = {"real": 300_001, "complex": 140_003, "float32": 600_007, "real_csc_update": 270_000}[case]
    A = powerlaw_matrix(n, 9, cplx=cplx, longest=3000).astype(T)
    if case == "real_csc_update":
        A = A.tocsc()
        A.sort_indices()
    rng = np.random.default_rng(12)
    b = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
    T64 = np.complex128 if cplx else np.float64
    A64, b64 = A.astype(T64), b.astype(T64)
    tol = 2e-5 if T == np.float32 else TOL
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    info = eu.host_pattern_info(A.tocsr(), T)
    assert info["sell_cut"] > 0                                    # irregular rows
    close(np.asarray(op.matvec(b)).astype(T64), A64 @ b64, 5e-6 if T == np.float32 else 1e-13, "column-blocked form %s: mul! vs scipy" % case)
    m = 16
    Ks = eu.arnoldi(op, b, m=m, ishermitian=False)
    Ko = ko.arnoldi(A64.tocsr(), b64, m=m, ishermitian=False)
    assert Ks.m == Ko.m
    close(np.asarray(Ks.getH()).astype(T64), Ko.getH(), tol, "column-blocked form %s: H of arnoldi! vs oracle" % case, mat=True)
    w = eu.expv(0.6, op, b, m=m, ishermitian=False)
    assert eu.expv.last_stats["path"] == ("two_kernel",) or list(eu.expv.last_stats["path"]) == ["two_kernel"]
    close(np.asarray(w).astype(T64), ko.expv(0.6, A64.tocsr(), b64, m=m, ishermitian=False), tol, "column-blocked form %s: expv vs oracle" % case)
    w1 = np.asarray(eu.expv(0.6, op, b, m=m, ishermitian=False))
    assert np.array_equal(np.asarray(w), w1)                       # fixed summation order: reproducible run to run
    if case == "real_csc_update":
        A2 = A.copy()
        A2.data = A.data * (1.0 + 0.2 * rng.random(A.nnz))
        op.update_values(A2)
        close(eu.expv(0.6, op, b, m=m, ishermitian=False), ko.expv(0.6, A2.tocsr(), b64, m=m, ishermitian=False), TOL,
              "column-blocked form after update_values (CSC entry order) vs oracle on the new matrix")


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["dia_c128", "dia_c64", "patch_grid_c128", "patch_grid_c64", "patch_band10_c128"])
def test_complex_windows_of_16_to_31_columns_on_the_single_pass_step(eu, form):
    """VERDICT r4 item 2 / missing 3: a complex operator with full Arnoldi at the default m = min(30, n) (arnoldi.jl:161-165,289-308;
    the reference's own GPU test is expv(t, A_gpu, b) on a ComplexF64 operator at default m, test/gpu/gputests.jl:41-58) ran all 30
    steps on the two-kernel step because the single-pass step took complex windows of <= 15 columns only.  The 24- and 32-column
    variants (two running sums per lane) take windows up to 31: the banded DIA form (n = 70 001: ragged last tile), the patch form
    of a 2-D grid, of a ten-offset band in its own ordering, ComplexF64 and ComplexF32; m = 20, 30, 31 (closing pass with a
    31-column window), 32 (no closing pass: tail kernels), an incomplete window of 20 columns over 40 steps; H, expv!, the whole
    call, overlapped == serial bit for bit."""
    rng = np.random.default_rng(59)
    T = np.complex64 if form.endswith("c64") else np.complex128
    tol = 2e-5 if T == np.complex64 else TOL
    if form.startswith("dia"):
        n = 70_001
        A0 = (c2_operator(n) * (1 + 0.25j)).tocsr()
    elif form.startswith("patch_grid"):
        k, rows = 180, 150
        n = k * rows
        A0 = (sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr") * (1 + 0.25j)).tocsr()
    else:
        n = 27_000
        offs = [-8, -6, -5, -3, -1, 0, 1, 2, 4, 7]
        A0 = sp.diags([(0.1 + 0.05 * rng.random(n - abs(o))) * (1 if o else -6.0) * (1 + 0.3j) for o in offs], offs, shape=(n, n), format="csr")
    A = A0.astype(T)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(T)
    A128, b128 = A.astype(np.complex128), b.astype(np.complex128)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    if form.startswith("patch"):
        assert op.patch_info["patch_form"], op.patch_info
    t = 0.4 - 0.2j
    for m, iop in ((20, 0), (30, 0), (31, 0), (32, 0), (40, 20)):
        ctx.set_pipeline_overlap(True)
        w = np.asarray(eu.expv(t, op, b, m=m, iop=iop, ishermitian=False)).copy()
        path = list(eu.expv.last_stats["path"])
        assert "pipeline" in path and ("patch" in path) == form.startswith("patch"), (form, m, iop, path)
        Ko = ko.KrylovSubspace(np.complex128, np.complex128, n, m)      # (the oracle once per case: expv = arnoldi! + expv!, krylov_phiv.jl:135-144)
        ko.arnoldi_(Ko, A128, b128, m=m, iop=iop, ishermitian=False)
        w_oracle = ko.expv_(np.empty(n, dtype=np.complex128), t, Ko)
        close(w.astype(np.complex128), w_oracle, tol if iop == 0 else 10 * tol,
              "%s m=%d iop=%d: expv vs oracle" % (form, m, iop))
        ctx.set_pipeline_overlap(False)
        w2 = np.asarray(eu.expv(t, op, b, m=m, iop=iop, ishermitian=False)).copy()
        assert np.array_equal(w, w2), "%s m=%d iop=%d: overlapped and serial forms differ" % (form, m, iop)
        ctx.set_pipeline_overlap(True)
        Ks = eu.KrylovSubspace(T, T, n, m, 0, ctx)
        eu.arnoldi_(Ks, op, b, m=m, iop=iop, ishermitian=False)
        assert Ks.m == Ko.m == m
        close(np.asarray(Ks.getH()).astype(np.complex128), Ko.getH(), tol, "%s m=%d iop=%d: H incl. H[m+1, m] vs oracle" % (form, m, iop), mat=True)
        close(np.asarray(eu.expv_(np.empty(n, dtype=T), t, Ks)).astype(np.complex128), w_oracle, tol,
              "%s m=%d iop=%d: expv! vs oracle" % (form, m, iop))
        if m == 30 and T == np.complex128:      # the basis itself (materialised from the raw columns + scales), first and last columns
            V = np.asarray(Ks.getV())
            Vo = Ko.getV()
            for c in (0, 17, 29, 30):
                close(V[:, c], Vo[:, c], 1e-10, "%s: basis column %d vs oracle" % (form, c))
    # arnoldi!(...; init = j): a continuation with a 25-column window picks the stored basis up on the single-pass step
    if T == np.complex128:
        Ks = eu.KrylovSubspace(T, T, n, 30, 0, ctx)
        eu.arnoldi_(Ks, op, b, m=22, ishermitian=False)
        eu.arnoldi_(Ks, op, b, m=30, init=22, ishermitian=False)
        Ko = ko.KrylovSubspace(np.complex128, np.complex128, n, 30)
        ko.arnoldi_(Ko, A128, b128, m=22, ishermitian=False)
        ko.arnoldi_(Ko, A128, b128, m=30, init=22, ishermitian=False)
        close(np.asarray(Ks.getH()), Ko.getH(), tol, "%s: continuation init = 22 -> m = 30: H vs oracle" % form, mat=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["shuffled_band", "shuffled_grid", "banded_ten_offsets", "complex_shuffled_grid", "shuffled_band_csc"])
def test_ordering_plan_cache_reuses_the_plan_of_a_pattern_seen_before(eu, kind):
    """operator creation works the row ordering / patch plan out from the PATTERN (0.4 .. 1.1 s at n = 1e6); a second
    operator with the same pattern takes the stored plan (expv_mi_plan_cache).  Same pattern + NEW values: a hit, identical storage
    decisions, results against the oracle for the new values, and bit-identical to an operator built with the cache switched off;
    a different pattern of the same size: a miss; capacity 0: never a hit."""
    rng = np.random.default_rng(71)
    cplx = kind.startswith("complex")
    csc = kind.endswith("_csc")          # (Julia's SparseMatrixCSC: the plan is made for the CSR form the library converts to, the value map composed)
    if kind in ("shuffled_band", "shuffled_band_csc"):
        n = 60_000
        A0 = c2_operator(n)
    elif kind in ("shuffled_grid", "complex_shuffled_grid"):
        k = 200
        n = k * 240
        A0 = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
        if cplx:
            A0 = (A0 * (1 + 0.25j)).tocsr()
    else:
        n = 40_000
        offs = [-8, -6, -5, -3, -1, 0, 1, 2, 4, 7]
        A0 = sp.diags([(0.1 + 0.05 * rng.random(n - abs(o))) * (1 if o else -6.0) for o in offs], offs, shape=(n, n), format="csr")
    if kind != "banded_ten_offsets":
        q = rng.permutation(n)
        A0 = A0[q][:, q].tocsr()
    A0.sort_indices()
    if csc:
        A0 = A0.tocsc()
        A0.sort_indices()
    b = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0)
    ctx = eu.Context()
    eu.plan_cache(clear=True, capacity=2)
    s0 = eu.plan_cache()
    op1 = eu.MIOperator(A0, ctx)
    s1 = eu.plan_cache()
    assert s1["misses"] == s0["misses"] + 1 and s1["hits"] == s0["hits"] and s1["plans"] == 1, (s0, s1)
    A1 = A0.copy()
    A1.data = A1.data * (1.0 + 0.1 * rng.random(A1.nnz))                    # same pattern, new values
    op2 = eu.MIOperator(A1, ctx)
    s2 = eu.plan_cache()
    assert s2["hits"] == s1["hits"] + 1 and s2["plans"] == 1, (s1, s2)
    assert op2.reorder_info["reordered"] == op1.reorder_info["reordered"] and op2.patch_info == op1.patch_info
    assert op2.reorder_info["bandwidth_after"] == op1.reorder_info["bandwidth_after"]
    m = 20
    w2 = np.asarray(eu.expv(0.5, op2, b, m=m, ishermitian=False)).copy()
    path2 = list(eu.expv.last_stats["path"])
    close(w2, ko.expv(0.5, A1.astype(np.complex128 if cplx else np.float64), b, m=m, ishermitian=False), TOL, "%s: operator from a cached plan vs oracle" % kind)
    close(np.asarray(op2 @ b), A1 @ b, 1e-13, "%s: mul! through the cached plan" % kind)
    # values-only update of an operator built from a cached plan scatters through the stored map
    op2.update_values(A0)
    close(np.asarray(eu.expv(0.5, op2, b, m=m, ishermitian=False)), ko.expv(0.5, A0.astype(np.complex128 if cplx else np.float64), b, m=m, ishermitian=False), TOL,
          "%s: update_values on an operator from a cached plan" % kind)
    # the same operator with the cache off: the same bits
    eu.plan_cache(clear=True, capacity=0)
    op3 = eu.MIOperator(A1, ctx)
    s3 = eu.plan_cache()
    assert s3["plans"] == 0 and s3["capacity"] == 0
    w3 = np.asarray(eu.expv(0.5, op3, b, m=m, ishermitian=False)).copy()
    assert list(eu.expv.last_stats["path"]) == path2
    assert np.array_equal(w2, w3), "%s: cached plan and fresh plan differ" % kind
    op4 = eu.MIOperator(A1, ctx)
    assert eu.plan_cache()["hits"] == s3["hits"]                            # capacity 0: nothing stored, nothing found
    # a different pattern of the same size and nnz count is a miss (compared entry by entry)
    eu.plan_cache(clear=True, capacity=2)
    op5 = eu.MIOperator(A1, ctx)
    B = A1.tocsr().copy().tolil()
    r = n // 3
    cols = B.rows[r]
    old_c = cols[0]
    new_c = (old_c + n // 2 + 11) % n
    assert new_c not in cols
    v = B[r, old_c]
    B[r, old_c] = 0
    B[r, new_c] = v
    B = B.tocsr()
    B.eliminate_zeros()
    B.sort_indices()
    assert B.nnz == A1.nnz and not np.array_equal(B.indices, A1.tocsr().indices)
    h0 = eu.plan_cache()["hits"]
    op6 = eu.MIOperator(B, ctx)
    assert eu.plan_cache()["hits"] == h0
    close(np.asarray(op6 @ b), B @ b, 1e-13, "%s: a changed pattern is analysed afresh" % kind)
    eu.plan_cache(clear=True, capacity=2)
    del op1, op2, op3, op4, op5, op6


@pytest.mark.gpu
def test_matrix_free_callback_sees_normalised_columns_by_default(eu):
    """the reference calls mul!(y, A, v_j) with |v_j| = 1 (arnoldi.jl:185, :306).  A callback that is only
    approximately linear -- a finite-difference Jacobian-vector product (f(u0 + eps v) - f(u0)) / eps with eps tuned for unit vectors, the
    usual matrix-free operator of an exponential integrator -- must see exactly that: the default (option matfree_fused = 0) hands it the
    normalised column and agrees with the oracle driving the SAME function; the two-kernel step (matfree_fused = 1, linear callbacks
    only) hands it beta_{j-1} v_j, and with |b| = 1e3 the truncation error of the difference quotient is 1e3 x larger: measurably worse."""
    import torch
    n, m = 600, 12
    rng = np.random.default_rng(606)
    A = rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)
    u0 = rng.standard_normal(n)
    b = rng.standard_normal(n)
    b *= 1.0e3 / np.linalg.norm(b)
    eps = 1.0e-7
    f_np = lambda u: A @ u + 0.5 * u * u * u
    J = A + np.diag(1.5 * u0 * u0)                      # the exact Jacobian at u0
    jv_np = lambda v: (f_np(u0 + eps * v) - f_np(u0)) / eps
    Ad, u0d = torch.as_tensor(A, device="cuda"), torch.as_tensor(u0, device="cuda")
    f_t = lambda u: Ad @ u + 0.5 * u * u * u
    f0 = f_t(u0d)
    jv_t = lambda v: (f_t(u0d + eps * v) - f0) / eps

    class FD:                                         # the oracle's operator: the same difference quotient on the host
        shape, dtype = (n, n), np.dtype(np.float64)
        def __matmul__(self, v): return jv_np(v)
    want = ko.expv(0.3, FD(), b, m=m, ishermitian=False)
    exact = ko.expv(0.3, J, b, m=m, ishermitian=False)
    errs = {}
    for fused in (0, 1):
        ctx = eu.Context()
        if fused:
            ctx.set_option("matfree_fused", 1)
        else:
            assert ctx.get_option("matfree_fused") == 0, "the default must be the reference's contract"
        op = eu.MIOperator(None, ctx, matvec=jv_t, shape=(n, n), dtype=np.float64, ishermitian=False)
        w = np.asarray(eu.expv(0.3, op, b, m=m, ishermitian=False))
        path = list(eu.expv.last_stats["path"])
        assert ("two_kernel" in path) == bool(fused), path
        errs[fused] = float(np.linalg.norm(w - exact) / np.linalg.norm(exact))
        if not fused:
            close(w, want, 1e-6, "finite-difference J*v callback, default path: expv vs the oracle driving the same callback")
    # the difference quotient's truncation error is O(eps |v|^2): unit columns 1e-7-ish, beta-scaled columns 1e3 x that and more
    assert errs[0] < 1e-5, errs
    assert errs[1] > 3 * errs[0], "scaled arguments should be visibly worse for this callback: %r" % (errs,)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.complex128, np.float32])
def test_deferred_closing_pass_is_collected_by_every_accessor(eu, T):
    """Round 6: arnoldi! / lanczos! with EXPV_MI_ARNOLDI_DEFER_TAIL (what the Python mirror and the Julia shim set) return when H[1:m, 1:m] is
    final; v_{m+1}, H[m+1, m] and the breakdown test of step m belong to the closing pass, which whoever touches the subspace next collects.
    Every accessor must give what the undeferred call gives, bit for bit: getH (incl. H[m+1, m]) first, Ks.m / wasbreakdown first, getV first,
    expv! first, resize! first, another factorisation first, destroy with the pass still pending; and a happy breakdown AT step m -- which only
    the closing pass can see -- is reported (Ks.m = m, wasbreakdown) like the oracle reports it (arnoldi.jl:370-374)."""
    rng = np.random.default_rng(83)
    n, m = 70_001, 12
    cplx = np.dtype(T).kind == "c"
    A = (c2_operator(n) * ((1 + 0.25j) if cplx else 1.0)).astype(T).tocsr()
    b = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(T)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)

    def fresh(defer):
        Ks = eu.KrylovSubspace(T, T, n, m + 4, 0, ctx)
        eu.arnoldi_(Ks, op, b, m=m, ishermitian=False, defer_tail=defer)
        return Ks
    K0 = fresh(False)
    # (expv! before getV combines the raw columns with scaled coefficients, after it the normalised columns: equal to rounding, not bit for bit)
    H0 = np.asarray(K0.getH()).copy()
    w0 = np.asarray(eu.expv_(np.empty(n, dtype=T), 0.4, K0)).copy()
    V0 = np.asarray(K0.getV()).copy()
    w0v = np.asarray(eu.expv_(np.empty(n, dtype=T), 0.4, K0)).copy()
    for first in ("getH", "m", "getV", "expv", "resize", "refactor", "set_m"):
        Ks = fresh(True)
        if first == "getH":
            assert np.array_equal(np.asarray(Ks.getH()), H0)
        elif first == "m":
            assert Ks.m == m and not Ks.wasbreakdown
        elif first == "getV":
            assert np.array_equal(np.asarray(Ks.getV()), V0)
        elif first == "expv":
            assert np.array_equal(np.asarray(eu.expv_(np.empty(n, dtype=T), 0.4, Ks)), w0)      # (its host exponential ran UNDER the closing pass)
        elif first == "resize":      # (resize! of a plain subspace starts it afresh, arnoldi.jl:80-93: the pending pass must be drained first, then a new factorisation)
            Ks.resize(m + 9)
            eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
        elif first == "refactor":
            eu.arnoldi_(Ks, op, b, m=m, ishermitian=False)
        elif first == "set_m":
            Ks.m = m
        assert np.array_equal(np.asarray(Ks.getH())[: m + 1, :m], H0[: m + 1, :m]), first
        assert np.array_equal(np.asarray(eu.expv_(np.empty(n, dtype=T), 0.4, Ks)), w0v if first == "getV" else w0), first
        assert np.array_equal(np.asarray(Ks.getV())[:, : m + 1], V0[:, : m + 1]), first
    Ks = fresh(True)
    del Ks                                                   # destroyed with the closing pass possibly still in flight
    ctx.sync()
    # a happy breakdown exactly at step m: b lives in an invariant subspace of dimension m of a block-diagonal operator
    if T != np.float32:
        k = 6
        blk = (rng.standard_normal((k, k)) + (1j * rng.standard_normal((k, k)) if cplx else 0)).astype(T)
        nn = 4096
        Ab = sp.block_diag([sp.csr_matrix(blk)] + [sp.identity(nn - k, dtype=T, format="csr") * T(-1.0)], format="csr").astype(T)
        bb = np.zeros(nn, dtype=T)
        bb[:k] = (rng.standard_normal(k) + (1j * rng.standard_normal(k) if cplx else 0)).astype(T)
        Ko = ko.arnoldi(Ab, bb, m=k, ishermitian=False)
        for defer in (False, True):
            Kd = eu.KrylovSubspace(T, T, nn, k, 0, ctx)
            eu.arnoldi_(Kd, Ab, bb, m=k, ishermitian=False, defer_tail=defer)
            assert (Kd.m, bool(Kd.wasbreakdown)) == (Ko.m, bool(Ko.wasbreakdown)), (defer, Kd.m, Kd.wasbreakdown, Ko.m, Ko.wasbreakdown)
            close(np.asarray(Kd.getH())[:k, :k], Ko.getH()[:k, :k], 1e-11, "breakdown at step m, defer=%s: H vs oracle" % defer, mat=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(513, 7), (2000, 30), (70_001, 30), (300_000, 40)])
def test_pipelined_lanczos_opt_in_mode(eu, n, m):
    """the opt-in `ortho = "pipelined"` mode of lanczos! (csrc/lanczos_pl.hip) -- NOT the reference's arithmetic: alpha_j,
    beta_j come from inner products by expansion, one pass behind (oracle/pipelined_lanczos.py: lanczos_p3 is the scheme in numpy).
    Its stated bars, against the REFERENCE recurrence (arnoldi.jl:388-403, 456-490) on the symmetric C2 operator: expv!(w, t, Ks) to 1e-11,
    H to 1e-11 of its largest entry, the basis orthonormal to 1e-9, beta and Ks.m equal; against the numpy restatement of the same scheme
    H to 1e-10 (same arithmetic, other summation order); the path flag says the mode ran; the default path is untouched (ortho = "auto"
    on the same subspace afterwards equals the oracle at the usual 1e-12)."""
    rng = np.random.default_rng(77)
    from oracle import pipelined_lanczos as pl
    A = c2_operator(n, sym=True)
    b = rng.standard_normal(n)
    ctx = eu.Context()
    op = eu.MIOperator(A, ctx)
    Ks = eu.KrylovSubspace(np.float64, np.float64, n, m, 0, ctx)
    eu.lanczos_(Ks, op, b, m=m, ortho="pipelined")
    Ko = ko.KrylovSubspace(np.float64, np.float64, n, m)
    ko.lanczos_(Ko, A, b, m=m)
    assert Ks.m == Ko.m == m and not Ks.wasbreakdown
    assert abs(Ks.beta - Ko.beta) <= 1e-13 * Ko.beta
    H, Ho = np.asarray(Ks.getH()), Ko.getH()
    close(H, Ho, 1e-11, "pipelined Lanczos n=%d m=%d: H vs the reference recurrence" % (n, m), mat=True)
    w = np.asarray(eu.expv_(np.empty(n), 0.7, Ks))
    close(w, ko.expv_(np.empty(n), 0.7, Ko), 1e-11, "pipelined Lanczos n=%d m=%d: expv! vs the reference recurrence" % (n, m))
    V = np.asarray(Ks.getV())[:, :m]
    assert float(np.abs(V.T @ V - np.eye(m)).max()) < 1e-9
    b0, al, be, _ = pl.lanczos_p3(A, b, m)
    assert np.abs(np.diag(H)[:m] - al).max() <= 1e-10 * np.abs(al).max() and np.abs(np.diag(H, -1)[:m] - be).max() <= 1e-10 * np.abs(be).max()
    wc = np.asarray(eu.expv(0.7, op, b, m=m, ishermitian=True, ortho="pipelined"))      # the whole-call form
    assert "pipelined_lanczos" in eu.expv.last_stats["path"], eu.expv.last_stats
    close(wc, w, 1e-13, "pipelined Lanczos: whole call vs lanczos! + expv!")
    eu.lanczos_(Ks, op, b, m=m)                                                          # the default path on the same storage
    close(np.asarray(Ks.getH()), Ho, TOL, "default Lanczos after a pipelined one on the same subspace", mat=True)
    # where the mode does not apply the call runs the default path and says so
    An = c2_operator(n)                                                                  # not Hermitian
    wn = np.asarray(eu.expv(0.7, eu.MIOperator(An, ctx), b, m=min(m, 30), ishermitian=False, ortho="pipelined"))
    assert "pipelined_lanczos" not in eu.expv.last_stats["path"]
    close(wn, ko.expv(0.7, An, b, m=min(m, 30), ishermitian=False), TOL, "ortho = pipelined on a non-Hermitian operator: the default path")


@pytest.mark.gpu
def test_ordering_plan_cache_keeps_element_types_of_equal_size_apart(eu):
    """the plan cache was keyed by sizeof(value), and Float64 / ComplexF32 are both 8 bytes -- but creation decides between
    orderings differently for real and complex types, so a ComplexF32 operator created first made a later Float64 operator with the
    same pattern inherit its plan: storage form and result bits depended on process history.  The key now holds the element type:
    same pattern, other type = a miss, and the operator comes out bit for bit as with the cache switched off."""
    rng = np.random.default_rng(72)
    k = 200
    n = k * 120
    A0 = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    q = rng.permutation(n)
    A0 = A0[q][:, q].tocsr()
    A0.sort_indices()
    b = rng.standard_normal(n)
    ctx = eu.Context()
    eu.plan_cache(clear=True, capacity=0)
    op_ref = eu.MIOperator(A0, ctx)
    w_ref = np.asarray(eu.expv(0.5, op_ref, b, m=12, ishermitian=False)).copy()
    strip = lambda d: {k: v for k, v in dict(d).items() if not k.endswith("_s")}      # (without the timing fields)
    info_ref = (strip(op_ref.reorder_info), strip(op_ref.patch_info))
    eu.plan_cache(clear=True, capacity=2)
    op_c = eu.MIOperator((A0 * (1 + 0.25j)).astype(np.complex64), ctx)          # 8-byte values, complex: its plan goes into the cache
    st0 = eu.plan_cache()
    op_r = eu.MIOperator(A0, ctx)                                              # 8-byte values, real, same pattern
    st1 = eu.plan_cache()
    assert st1["hits"] == st0["hits"], (st0, st1)                              # ... must NOT take the complex plan
    assert (strip(op_r.reorder_info), strip(op_r.patch_info)) == info_ref
    assert np.array_equal(np.asarray(eu.expv(0.5, op_r, b, m=12, ishermitian=False)), w_ref)
    op_r2 = eu.MIOperator(A0, ctx)                                             # the same type again: a hit, same bits
    assert eu.plan_cache()["hits"] == st1["hits"] + 1
    assert np.array_equal(np.asarray(eu.expv(0.5, op_r2, b, m=12, ishermitian=False)), w_ref)
    eu.plan_cache(clear=True, capacity=2)
    del op_c, op_r, op_r2, op_ref


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,iop,herm", [(1, 22, 0, False), (17, 2, 7, False), (129, 6, 7, True), (130, 30, 0, False), (257, 12, 2, False), (1001, 30, 0, True)])
def test_matrix_free_operator_on_the_two_kernel_step_odd_sizes(eu, n, m, iop, herm):
    """Round 5: matrix-free operators (docs/src/interfaces.md:7-36, basictests.jl:786-816) run the two-kernel step, their mul! feeding its
    first kernel.  Found by tests/fuzz_parity.py (seed 5055 cases 8940 / 9732 / 13542: n = 1, 129, 17): the kernel reads y~ in whole
    16-byte packs, the callback writes n elements -- for odd n the pack's last element was whatever the buffer held.  The buffer is now
    the library's own, zero-filled.  Odd and even n, m > n (happy breakdown on the first step), Lanczos, an incomplete window, and the
    modular path (option matfree_fused = 0) beside it."""
    import torch
    rng = np.random.default_rng([5055, n])
    A = rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)
    if herm:
        A = (A + A.T) * 0.5
    b = rng.standard_normal(n)
    Ad = torch.as_tensor(A, device="cuda")
    want = ko.expv(0.7, A, b, m=m, iop=iop, ishermitian=herm)
    for fused in (1, 0):
        ctx = eu.Context()
        ctx.set_option("matfree_fused", fused)
        # (poison the allocator's free blocks: a stale buffer must not be able to help)
        junk = torch.full((4 * (n + 512),), float("nan"), dtype=torch.float64, device="cuda")
        del junk
        op = eu.MIOperator(None, ctx, matvec=lambda x: Ad @ x, shape=(n, n), dtype=np.float64, ishermitian=herm)
        for rep in range(2):
            w = np.asarray(eu.expv(0.7, op, b, m=m, iop=iop, ishermitian=herm))
            path = list(eu.expv.last_stats["path"])
            assert ("two_kernel" in path) == bool(fused) or "modular" in path, path
            close(w, want, 1e-10, "matrix-free n=%d m=%d iop=%d herm=%d fused=%d rep %d: expv vs oracle" % (n, m, iop, herm, fused, rep))
        Ks = eu.arnoldi(op, b, m=min(m, 30), iop=iop, ishermitian=herm)
        Ko = ko.arnoldi(A, b, m=min(m, 30), iop=iop, ishermitian=herm)
        assert Ks.m == Ko.m and bool(Ks.wasbreakdown) == bool(Ko.wasbreakdown), (Ks.m, Ko.m)
        k = Ks.m
        close(np.asarray(Ks.getH())[:k, :k], np.asarray(Ko.getH())[:k, :k], 1e-10, "matrix-free n=%d fused=%d: H vs oracle" % (n, fused), mat=True)

"""SURVEY 8(f) row f3, closed by measurement: can the stopping test of expv(...; mode = :error_estimate) be updated in O(j) per step?

The test needs  |e_j' exp(t T_j) e_1|  for every j (krylov_phiv_error_estimate.jl:58-68, :174-200: sigma = beta_j * beta_0 * |v_j|).
The library computes it from two eigenvector rows of T_j: O(j^2) per step, bit-identical to the reference's full eigen-decomposition.
The only O(j) route is a partial-fraction form  exp(x) ~ sum_k c_k / (z_k - x):  for a tridiagonal T

    e_j' (z I - t T_j)^-1 e_1 = t^(j-1) beta_1 ... beta_(j-1) / det(z I - t T_j),      det_j = (z - t alpha_j) det_(j-1) - t^2 beta_(j-1)^2 det_(j-2)

so every pole costs O(1) per Lanczos step.  The poles must suit the set t * spectrum(A):
  (a) t real, A negative semi-definite (diffusion): ONE fixed set of poles works for every A (Trefethen-Weideman-Schmelzer parabolic
      contour, error ~ 2.85^-N on the whole negative axis) -- the recurrence reproduces the stopping step.
  (b) t imaginary (Schroedinger -- the reference's OWN test of the mode, test/basictests.jl:756-784, is expv(-im, dt*A, b)): exp(i x) on
      [-R, R] needs a degree that grows with R = |t| ||A||, the negative-axis poles give garbage (the arguments lie outside the contour),
      and a contour wide enough for R costs ~R + log(1/eps) poles: O(R) per step instead of O(j^2) with j <= 30 -- no gain, and the
      decisions are no longer bit-identical to the eigen-decomposition.
Result (printed below): (a) same stopping step; (b) entries wrong by O(1), and at three times the reference test's time step the
recurrence stops at step 1 with a wrong result.  Row f3 stays "O(j^2); O(j) infeasible at parity".
"""
import numpy as np


def lanczos(A, b, m):
    n = len(b)
    V = np.zeros((n, m + 1), dtype=complex)
    al, be = np.zeros(m), np.zeros(m)
    V[:, 0] = b / np.linalg.norm(b)
    for j in range(m):
        y = A @ V[:, j]
        al[j] = np.real(np.vdot(V[:, j], y))
        y -= al[j] * V[:, j] + (be[j - 1] * V[:, j - 1] if j else 0)
        be[j] = np.linalg.norm(y)
        V[:, j + 1] = y / be[j]
    return al, be


def last_entry_eig(al, be, t, j):                      # what the library / the reference compute: e_j' exp(t T_j) e_1
    T = np.diag(al[:j]) + np.diag(be[:j - 1], 1) + np.diag(be[:j - 1], -1)
    lam, Q = np.linalg.eigh(T)
    return (Q[j - 1] * np.exp(t * lam)) @ Q[0]


def parabola_poles(N):                                 # Weideman & Trefethen 2007, parabolic contour, N midpoint nodes
    th = np.pi * (2 * np.arange(N) + 1 - N) / N
    z = N * (0.1309 - 0.1194 * th ** 2 + 0.2500j * th)
    w = N * (-0.1194 * 2 * th + 0.2500j)
    return z, (-1j / N) * np.exp(z) * w                # exp(x) ~ sum_k c_k / (z_k - x)   (x on the negative real axis)


def last_entry_recurrence(al, be, t, m, z, c):
    """all j = 1..m in ONE sweep: O(len(z)) per step."""
    out = np.zeros(m, dtype=complex)
    d2, d1 = np.ones_like(z), z - t * al[0]
    prod = 1.0 + 0j
    out[0] = np.sum(c / d1)
    for j in range(2, m + 1):
        d2, d1 = d1, (z - t * al[j - 1]) * d1 - (t * be[j - 2]) ** 2 * d2
        prod = prod * t * be[j - 2]
        out[j - 1] = np.sum(c * prod / d1)
    return out


def stopping_step(vals, be, beta0, eps):
    for j, v in enumerate(vals, 1):
        if be[j - 1] * beta0 * abs(v) < eps:
            return j
    return len(vals)


rng = np.random.default_rng(0)
n, m = 300, 30
z, c = parabola_poles(32)
print("case                                   stopping step (eig)   (recurrence)   max |eig - recurrence| over j")
# (a) diffusion: t = 1, A = -(random SPD), spectrum on the negative axis
M = rng.random((n, n)); S = -(M @ M.T) / n
b = rng.random(n)
al, be = lanczos(S, b.astype(complex), m)
ref = np.array([last_entry_eig(al, be, 0.1, j) for j in range(1, m + 1)])
rec = last_entry_recurrence(al, be, 0.1, m, z, c)
eps = 1e-10 + 1e-10 * np.linalg.norm(b)
print("(a) exp(0.1 A) b, A negative definite      %3d                %3d            %.2e" %
      (stopping_step(ref, be, np.linalg.norm(b), eps), stopping_step(rec, be, np.linalg.norm(b), eps), np.max(np.abs(ref - rec))))
# (b) the reference's own test of the mode: expv(-im, dt * A, b), A = Hermitian(rand(n, n)), dt = 0.1 -- and the same with 3 dt
for dt in (0.1, 0.3):
    H = rng.random((n, n)); H = (H + H.T) / 2
    bc = rng.random(n) + 1j * rng.random(n)
    al, be = lanczos(dt * H, bc, m)
    ref = np.array([last_entry_eig(al, be, -1j, j) for j in range(1, m + 1)])
    rec = last_entry_recurrence(al, be, -1j, m, z, c)
    eps = 1e-10 + 1e-10 * np.linalg.norm(bc)
    s_ref, s_rec = stopping_step(ref, be, np.linalg.norm(bc), eps), stopping_step(rec, be, np.linalg.norm(bc), eps)
    print("(b) exp(-i %.1f A) b, A = Hermitian(rand)   %3d                %3d            %.2e   %s" %
          (dt, s_ref, s_rec, np.max(np.abs(ref - rec)), "<- stopping step lost" if s_ref != s_rec else "<- entries of the first steps wrong by O(1)"))
    print("    |t| ||A|| = %.1f: the arguments -i t lambda lie outside the contour, which crosses the imaginary axis at +-%.1f i;"
          " |e_1' exp(t T_1) e_1| = %.3f (eig) / %.3g (recurrence)" %
          (np.max(np.abs(np.linalg.eigvalsh(dt * H))), 32 * 0.25 * np.sqrt(0.1309 / 0.1194), abs(ref[0]), abs(rec[0])))
print("A contour wide enough for R = |t| ||A|| needs ~R + log(1/eps) poles: O(R) work per step against O(j^2), j <= 30, for the two\n"
      "eigenvector rows -- no gain, and decisions that are no longer bit-identical to the reference's eigen-decomposition.")

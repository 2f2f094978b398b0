"""What would a pipelined Lanczos recurrence cost in accuracy?  (CPU only, numpy)

Runs the reference recurrence (oracle/pipelined_lanczos.py: lanczos_ref = arnoldi.jl:388-403), the Ghysels-Vanroose
one-reduction form (lanczos_p1) and the form whose scalars arrive one pass late (lanczos_p2: inner products by expansion from a
Gram matrix -- what a device pass that never waits for the previous pass' reduction needs) on

  * the symmetric C2 operator (bench.py: offsets -2..2, symmetrised), n = 2e3 and 2e5, t = 1, m = 30
  * the complex Hermitian tridiagonal operator of /root/reference/test/basictests.jl:731-754 (p = -im * Tridiagonal(-e, 0, e),
    complex vector; imaginary time)
  * rand(300, 300) made Hermitian (basictests.jl:756-784), t = 1, m = 30

and prints  |w_pipe - w_ref| / |w_ref|,  max |H_pipe - H_ref| / max |H_ref|,  loss of orthogonality max |V'V - I|  of each,
plus each form's distance to the dense truth exp(tA)b where n allows.

    python tools/pipelined_lanczos_accuracy.py > profiles/r06_pipelined_lanczos_accuracy.txt
"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pipelined_lanczos as pl  # noqa: E402


def c2_symmetric(n):
    offs = (-2, -1, 0, 1, 2)
    vals = (0.3, 1.2, -2.0, 1.2, 0.3)
    return sp.diags([np.full(n - abs(o), v) for o, v in zip(offs, vals)], offs, format="csr")


def c2_symmetric_variable(n, rng):
    """the same pattern with random symmetric coefficients (the constant stencil has a very regular spectrum)"""
    d0 = -2.0 + 0.5 * rng.standard_normal(n)
    d1 = 1.2 + 0.3 * rng.standard_normal(n - 1)
    d2 = 0.3 + 0.1 * rng.standard_normal(n - 2)
    return sp.diags([d2, d1, d0, d1, d2], (-2, -1, 0, 1, 2), format="csr")


def hermitian_tridiagonal(rng, n=100):
    """basictests.jl:731-754: p = -im * Tridiagonal(-e, 0e, e) (a momentum operator: complex Hermitian), v = rand(ComplexF64, n)"""
    e = np.ones(n - 1)
    H = (-1j * sp.diags([-e, np.zeros(n), e], (-1, 0, 1))).tocsr()
    return H, rng.random(n) + 1j * rng.random(n)


def report(name, A, b, t, m, dense_truth=True):
    ref = pl.lanczos_ref(A, b, m)
    w_ref = pl.expv_from_lanczos(t, *ref, m)
    Hmax = max(np.abs(ref[1]).max(), np.abs(ref[2]).max())
    truth = None
    if dense_truth and A.shape[0] <= 4000:
        Ad = A.toarray() if sp.issparse(A) else A
        truth = sla.expm(t * Ad) @ b
    print("%s   n=%d  m=%d  t=%s" % (name, A.shape[0], m, t))
    rows = [("reference recurrence (arnoldi.jl:388-403)", ref, w_ref)]
    for label, fn in (("p1: Ghysels-Vanroose, one reduction per step", pl.lanczos_p1),
                      ("p2: scalars one pass late (Gram expansion)", pl.lanczos_p2),
                      ("p3: the device scheme (recomputed A v, A^2 v)", pl.lanczos_p3)):
        r = fn(A, b, m)
        rows.append((label, r, pl.expv_from_lanczos(t, *r, m)))
    for label, r, w in rows:
        V = r[3][:, :m]
        G = V.conj().T @ V
        loss = np.abs(G - np.eye(m)).max()
        dH = max(np.abs(r[1] - ref[1]).max(), np.abs(r[2] - ref[2]).max()) / Hmax
        dw = np.linalg.norm(w - w_ref) / np.linalg.norm(w_ref)
        line = "   %-48s |w - w_ref|/|w_ref| = %8.2e   max|H - H_ref|/max|H| = %8.2e   max|V'V - I| = %8.2e" % (label, dw, dH, loss)
        if truth is not None:
            line += "   |w - exp(tA)b|/|.| = %8.2e" % (np.linalg.norm(w - truth) / np.linalg.norm(truth))
        print(line)
    print()


def main():
    rng = np.random.default_rng(2026)
    for n in (2000, 200000):
        A = c2_symmetric(n)
        report("symmetric C2 operator (constant coefficients)", A, rng.standard_normal(n), 1.0, 30)
        A = c2_symmetric_variable(n, rng)
        report("symmetric C2 pattern, random coefficients", A, rng.standard_normal(n), 1.0, 30)
    H, psi = hermitian_tridiagonal(rng)
    report("complex Hermitian tridiagonal (basictests.jl:731-754), m = 15 as there, t = -1i", H, psi, -1.0j, 15)
    report("the same, m = 30", H, psi, -1.0j, 30)
    M = rng.random((300, 300))
    A = (M + M.T) / 2
    report("rand(300,300) Hermitian (basictests.jl:756-784)", A, rng.random(300), 1.0, 30)
    report("the same, m = 15", A, rng.random(300), 1.0, 15)
    # a longer run: what the recurrences do once Ritz values have converged (loss of orthogonality sets in)
    A = c2_symmetric_variable(2000, rng)
    report("symmetric C2 pattern, random coefficients, m = 60", A, rng.standard_normal(2000), 1.0, 60)
    print("Traffic per step at n = 1e6, fp64, 5 diagonals in DIA form (40 MB): reference recurrence on the single-pass step 40 + 8*(2 + 1 + 1 + 1) = 80 MB;\n"
          "p1: v_j, v_{j-1}, z_j, z_{j-1}, q_{j-1} read, v_{j+1}, z_{j+1}, q_j written + the operator = 40 + 8*8 = 104 MB (1.3 x);\n"
          "p2: v, z, q and their predecessors read, three written + the operator = 40 + 8*9 = 112 MB (1.4 x).\n"
          "A p1 pass still needs alpha_j, beta_j before it can form z_{j+1} -- it shortens the chain by the operator phase only; only p2 takes the\n"
          "reduction chain off the critical path.")


if __name__ == "__main__":
    main()

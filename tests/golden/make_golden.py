"""Generates the golden fixtures under tests/golden/ (run in the build container:
`python tests/golden/make_golden.py`).  A fixture is DATA: seeded inputs (or the recipe to rebuild
them deterministically) and expected outputs -- dense scipy truths and the oracle's H / w.  The
reference's own tests hold no stored vectors (they compare against `exp(t*A)*b` in-process with
Julia's RNG), so the expected values come from scipy.linalg.expm here and from the oracle, which is
itself pinned to the reference's deterministic KATs (tests/test_oracle_kat.py)."""
import os
import sys

import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import krylov_oracle as ko  # noqa: E402
from tests._util import c2_operator, dense_phis, mkA  # noqa: E402


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(name, os.path.getsize(path), "bytes")


# 1. BASELINE config 2 pattern at n = 2000 (non-symmetric 5-diagonal, Arnoldi): H, beta, w, dense truth
n, m, t = 2000, 30, 1.0
A = c2_operator(n)
b = np.random.default_rng(3).standard_normal(n)
Ks = ko.arnoldi(A, b, m=m, ishermitian=False)
w = ko.expv_(np.empty(n), t, Ks)
save("c2_n2000_m30", n=n, m=m, t=t, b_seed=3, H=Ks.getH(), beta=Ks.beta, w=w, w_dense=sl.expm(t * A.toarray()) @ b)

# 2. symmetric variant (Lanczos)
As = c2_operator(n, sym=True)
KsL = ko.arnoldi(As, b, m=m)
save("c2sym_n2000_m30", n=n, m=m, t=t, b_seed=3, H=KsL.getH(), beta=KsL.beta, w=ko.expv_(np.empty(n), t, KsL),
     w_dense=sl.expm(t * As.toarray()) @ b)

# 3. BASELINE config 1: dense 512 x 512, t = 1 (A, v from seeds 1, 2)
A1 = np.random.default_rng(1).standard_normal((512, 512)) / np.sqrt(512)
v1 = np.random.default_rng(2).standard_normal(512)
K1 = ko.arnoldi(A1, v1, m=30)
save("c1_dense512", n=512, m=30, t=1.0, H=K1.getH(), beta=K1.beta, w=ko.expv_(np.empty(512), 1.0, K1),
     w_dense=sl.expm(A1) @ v1)

# 4. the reference's deterministic phiv KAT (basictests.jl:569-573) and mkA / b = 1/i inputs (:859-868)
n3 = 30
A3 = np.diag(np.ones(n3 - 1), -1) + 30 * np.eye(n3) + np.diag(np.ones(n3 - 1), 1)
save("kat_phiv_tridiag30", Q=ko.phiv(0.1, A3, np.ones(n3), 10),
     phi1=np.linalg.solve(0.1 * A3, (sl.expm(0.1 * A3) - np.eye(n3)) @ np.ones(n3)))
n4 = 64
A4, b4 = mkA(n4), 1.0 / np.arange(1, n4 + 1)
save("kat_mkA64", W=ko.phiv(0.1, A4, b4, 3, m=30), W_dense=np.stack([P @ b4 for P in dense_phis(0.1 * A4, 3)], axis=1),
     w=ko.expv(0.1, A4, b4, m=30))

# 5. adaptive phiv_timestep on the reference's operator (basictests.jl:666-682), own seeded B
n5, K, t5 = 100, 4, 5.0
A5 = sp.diags([np.ones(n5 - 1), -2 * np.ones(n5), np.ones(n5 - 1)], [-1, 0, 1], format="csc")
B5 = np.random.default_rng(14).standard_normal((n5, K + 1))
st = {}
U5 = ko.phiv_timestep(np.array([t5 / 2, t5]), A5, B5, adaptive=True, tol=1e-7, stats=st)
Ph, Phh = dense_phis(t5 * A5.toarray(), K), dense_phis(t5 / 2 * A5.toarray(), K)
save("adaptive_phiv_timestep", U=U5, num_timesteps=st["num_timesteps"], matvecs=st["matvecs"], m_final=st["m"],
     u_exact=sum(t5 ** i * Ph[i] @ B5[:, i] for i in range(K + 1)),
     uhalf_exact=sum((t5 / 2) ** i * Phh[i] @ B5[:, i] for i in range(K + 1)))

# 6. kiops: real (reference behaviour) and the complex extension (config 4 pattern, no reference behaviour)
n6 = 400
A6 = c2_operator(n6).tocsc()
u6 = np.random.default_rng(11).standard_normal((n6, 3))
w6, s6 = ko.kiops(1.0, A6, u6)
A6c = (c2_operator(300) * (1 + 0.25j)).tocsc()
u6c = np.random.default_rng(6).standard_normal(300) + 1j * np.random.default_rng(60).standard_normal(300)
w6c, s6c = ko.kiops(1.0, A6c, u6c, allow_complex=True, ishermitian=False)
save("kiops", w=w6, stats=np.array(s6), wc=w6c, statsc=np.array(s6c), wc_dense=sl.expm(A6c.toarray()) @ u6c)

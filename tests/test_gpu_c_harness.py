"""A plain-C program that calls libexpv_mi.so the way julia/MIKrylov.jl does (1-based Int64 CSC, column-major, structs by
reference, device vectors from expv_mi_malloc, callback trampoline, default complete-on-return outputs, create-use-destroy of
subspaces) -- the closest thing to running the Julia shim this image allows.  The expected values come from
the oracle; the harness is compiled with gcc against include/expv_mi.h."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "exponentialutilities.jl_amd")


def build_harness(out_dir):
    exe = os.path.join(out_dir, "abi_harness")
    cmd = [shutil.which("gcc") or "gcc", "-O1", "-Wall", "-Wextra", "-std=c11", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_harness", "abi_harness.c"), "-o", exe, "-L", PKG, "-lexpv_mi", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_harness_compiles_against_the_header(tmp_path):
    """CPU-side: include/expv_mi.h is valid C11 and every call of the harness matches a prototype (-Wall -Wextra clean)."""
    exe = build_harness(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_harness_runs_the_shims_call_sequence(tmp_path):
    n, m, k, K = 20_000, 25, 3, 3
    rng = np.random.default_rng(17)
    d = [0.3 + 0.1 * rng.random(n - 2), 1.2 + 0.1 * rng.random(n - 1), -2.0 + 0.1 * rng.random(n), 0.8 + 0.1 * rng.random(n - 1),
         -0.1 + 0.1 * rng.random(n - 2)]
    A = sp.diags(d, [-2, -1, 0, 1, 2], format="csc")
    A.sort_indices()
    As = A.copy()
    S = ((A + A.T) * 0.5).tocsc()
    S.sort_indices()
    assert np.array_equal(S.indptr, A.indptr) and np.array_equal(S.indices, A.indices)      # same pattern: only the values differ
    b = rng.standard_normal(n)
    B = np.asfortranarray(rng.standard_normal((n, K)))
    ts = np.array([0.3, 0.7])
    Ko = ko.arnoldi(A, b, m=m, ishermitian=False)
    w = ko.expv_(np.empty(n), 0.7, Ko)
    Wp = np.asfortranarray(ko.phiv_(np.empty((n, k + 1), order="F"), 0.7, Ko, k))
    wl = ko.expv(0.7, S, b, m=m, ishermitian=True)
    st = {}
    U = np.asfortranarray(ko.phiv_timestep(ts.copy(), A, B, adaptive=True, tol=1e-8, stats=st))
    wk, sk = ko.kiops(0.7, A, B)
    d = str(tmp_path)
    np.array([n, A.nnz, m, k, K, 0], dtype=np.int64).tofile(os.path.join(d, "meta.i64"))
    (A.indptr.astype(np.int64) + 1).tofile(os.path.join(d, "colptr.i64"))            # Julia's 1-based SparseMatrixCSC
    (A.indices.astype(np.int64) + 1).tofile(os.path.join(d, "rowval.i64"))
    A.data.astype(np.float64).tofile(os.path.join(d, "nzval.f64"))
    S.data.astype(np.float64).tofile(os.path.join(d, "nzsym.f64"))
    b.tofile(os.path.join(d, "b.f64"))
    w.tofile(os.path.join(d, "w_expv.f64"))
    np.asfortranarray(Ko.getH()[: m + 1, :m]).ravel(order="F").tofile(os.path.join(d, "H.f64"))
    Wp.ravel(order="F").tofile(os.path.join(d, "W_phiv.f64"))
    np.asarray(wl).tofile(os.path.join(d, "w_lanczos.f64"))
    B.ravel(order="F").tofile(os.path.join(d, "B.f64"))
    U.ravel(order="F").tofile(os.path.join(d, "U_timestep.f64"))
    ts.tofile(os.path.join(d, "ts.f64"))
    np.asarray(wk).ravel(order="F").tofile(os.path.join(d, "w_kiops.f64"))
    np.array([st["num_timesteps"], st["matvecs"], st["m"]] + [int(v) for v in sk], dtype=np.int64).tofile(os.path.join(d, "stats.i64"))
    exe = build_harness(d)
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    print(r.stderr)
    lines = [l for l in r.stdout.splitlines() if l.startswith("CHECK")]
    assert r.returncode == 0 and "HARNESS OK" in r.stdout, r.stdout + r.stderr
    assert len(lines) >= 20 and all(l.endswith("OK") for l in lines)
    try:
        from tests._util import close
        for l in lines:                       # the measured errors join the parity report
            name = l[6: l.index(" err=")]
            err = float(l.split("err=")[1].split()[0])
            bar = float(l.split("bar=")[1].split()[0])
            if bar > 0:
                close(err, 0.0, bar, "C harness (shim call sequence): " + name, absolute=True)
    except ImportError:
        pass

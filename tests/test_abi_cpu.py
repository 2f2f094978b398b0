"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol that
include/expv_mi.h declares, its host small-dense functions agree with the oracle, and the GPU path
fails loudly (no silent CPU fallback).  No compute calls that need a GPU."""
import os
import re

import numpy as np
import pytest
import scipy.linalg as sl

import expv_mi_loader
from oracle import krylov_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eu():
    if not os.path.exists(os.path.join(ROOT, "exponentialutilities.jl_amd", "libexpv_mi.so")):
        expv_mi_loader.build()
    return expv_mi_loader.load()


def test_every_declared_symbol_is_exported_and_bound(eu):
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "expv_mi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(expv_mi_[a-z0-9_]+)\s*\(", hdr, flags=re.I))
    declared -= {"expv_mi_matvec_fn", "expv_mi_print_fn"}
    lib = ctypes.CDLL(os.path.join(ROOT, "exponentialutilities.jl_amd", "libexpv_mi.so"))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in expv_mi.h but not exported: {missing}"
    from exponentialutilities_jl_amd import _lib
    unbound = sorted(declared - set(_lib.PROTOTYPES))
    assert not unbound, f"declared but not bound in _lib.PROTOTYPES: {unbound}"
    assert len(declared) >= 45


def test_no_cpu_fallback_without_gpu(eu):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(eu.ExpvMIError) as ei:
        eu.Context()
    assert ei.value.kind == "HIPError"
    with pytest.raises(eu.ExpvMIError):
        eu.expv(1.0, np.eye(4), np.ones(4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "exponentialutilities.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


@pytest.mark.parametrize("T", [float, complex])
@pytest.mark.parametrize("scale", [30.0, 3.0, 1.5, 0.5, 0.1, 0.005])
def test_host_expm_every_pade_branch(eu, T, scale):
    """basictests.jl:952-974 design, on the product's own host Higham-2005 routine."""
    rng = np.random.default_rng(7)
    n = 40
    A0 = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if T is complex else 0)
    A = scale * A0 / np.linalg.norm(A0, 1)
    E = eu.host_expm(A)
    assert np.linalg.norm(E - sl.expm(A)) / np.linalg.norm(sl.expm(A)) < 1e-11
    assert np.linalg.norm(E - ko.exponential_(A)) / np.linalg.norm(E) < 1e-13


def test_host_expm_balancing(eu):
    rng = np.random.default_rng(3)
    A = np.triu(rng.standard_normal((8, 8))) * np.logspace(-3, 3, 8)[:, None] * 1e-2
    A[3, 0] = 1.0
    A[6, 2] = 3.0
    assert np.linalg.norm(eu.host_expm(A) - ko.exponential_(A)) / np.linalg.norm(ko.exponential_(A)) < 1e-13


@pytest.mark.parametrize("t", [0.3, -1.7, 0.2 - 0.7j])
@pytest.mark.parametrize("n", [1, 2, 12, 30])
def test_host_symtridiag_expcol(eu, t, n):
    rng = np.random.default_rng(n)
    d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 0))
    M = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.abs(eu.host_symtridiag_expcol(d, e, t) - sl.expm(t * M)[:, 0]).max() < 1e-13


@pytest.mark.parametrize("T", [float, complex])
def test_host_phiv_dense(eu, T):
    rng = np.random.default_rng(11)
    A = rng.standard_normal((9, 9)) + (1j * rng.standard_normal((9, 9)) if T is complex else 0)
    v = rng.standard_normal(9).astype(T)
    w = np.empty((9, 5), dtype=T, order="F")
    ko.phiv_dense_(w, A, v, 4)
    assert np.abs(eu.host_phiv_dense(A, v, 4) - w).max() < 1e-13


def test_host_expm_rejects_nan(eu):
    """Balancing never terminates on NaN input (LAPACK.gebal! guards with chkfinite): must raise, not hang."""
    with pytest.raises(eu.ExpvMIError):
        eu.host_expm(np.full((3, 3), np.nan))
    with pytest.raises(ValueError):
        ko.exponential_(np.full((3, 3), np.nan))

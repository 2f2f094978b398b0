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


def test_host_pattern_info_selects_the_storage_forms(eu):
    """The format / path decision of operator creation is host logic: check it without a GPU."""
    import scipy.sparse as sp
    n = 5000
    c2 = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csr")
    i = eu.host_pattern_info(c2)
    assert i["sell"] and i["bandwidth"] == 2 and i["pipeline_dia_diagonals"] == 5 and i["general_dia_diagonals"] == 0
    assert i["path"].startswith("pipeline, halo")
    nine = sp.diags([1.0] * 9, list(range(-4, 5)), shape=(n, n), format="csr")          # > 8 offsets: SELL slots, still banded
    i = eu.host_pattern_info(nine)
    assert i["pipeline_dia_diagonals"] == 0 and i["bandwidth"] == 4 and i["path"].startswith("pipeline, halo")
    grid = sp.diags([1.0, 1.0, -4.0, 1.0, 1.0], [-70, -1, 0, 1, 70], shape=(n, n), format="csr")
    i = eu.host_pattern_info(grid)
    assert i["pipeline_dia_diagonals"] == 0 and i["general_dia_diagonals"] == 5 and i["general_dia_max_offset"] == 70
    assert "wave" in i["path"]
    assert eu.host_pattern_info(grid, np.complex128)["path"] == "two-kernel step"          # complex: general DIA, no pipeline
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(n), 5)
    cols = np.clip(rows + rng.integers(-300, 301, size=rows.size), 0, n - 1)
    irr = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n))
    irr.sum_duplicates()
    i = eu.host_pattern_info(irr)
    assert i["sell"] and i["general_dia_diagonals"] == 0 and 0 < i["sell_wave_reach"] <= 300 + 511
    ragged = sp.lil_matrix((n, n))
    ragged[0, :400] = 1.0                                                                  # one long row per slice-full of short ones
    ragged.setdiag(2.0)
    i = eu.host_pattern_info(ragged.tocsr())
    assert i["path"].startswith("modular") or i["sell"]                                    # padding rule decides; never crashes
    unsorted = sp.csr_matrix((np.array([1.0, 2.0, 3.0]), np.array([1, 0, 1]), np.array([0, 2, 3])), shape=(2, 2))
    unsorted.has_sorted_indices = True                                                     # keep scipy from sorting them
    assert eu.host_pattern_info(unsorted)["rows_sorted_unique"] in (True, False)


def test_operator_fingerprint_detects_in_place_changes(eu):
    """an implicitly uploaded host matrix is reused only while its content fingerprint is unchanged."""
    import scipy.sparse as sp
    from exponentialutilities_jl_amd import api
    A = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(2000, 2000), format="csr")
    f0 = api._fingerprint(A)
    assert api._fingerprint(A) == f0
    A.data[1234] += 1e-13
    f1 = api._fingerprint(A)
    assert f1 != f0
    A *= 0.5                      # rebinds A.data in scipy: pointer and checksum change
    assert api._fingerprint(A) != f1
    A.indices[7] = 9
    assert api._fingerprint(A) != f1
    D = np.arange(12.0).reshape(3, 4)[:, :3].copy()
    g0 = api._fingerprint(D)
    D[2, 1] = -D[2, 1]
    assert api._fingerprint(D) != g0
    assert api._fingerprint(np.zeros((3, 3), dtype=complex)) != api._fingerprint(np.zeros((3, 3)))

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


def test_inline_assembly_wide_stores_are_padded_against_the_store_data_hazard():
    """gfx940+: a VMEM store of more than 8 bytes reads its data registers late; a VALU write to one of them within 2 wait states of the
    store reaches memory instead of the stored value.  The compiler pads its own stores (GCNHazardRecognizer) but not inline assembly --
    round 6 found `global_store_dwordx4 v[104:105], v[80:83] ... ; v_mul_f32 v80, ...` in the overlapped Float32 SELL wave form.  Every
    hand-written store wider than 8 bytes must carry its own `s_nop 1`."""
    import re
    pkg = os.path.join(ROOT, "exponentialutilities.jl_amd", "csrc")
    seen = 0
    for f in sorted(os.listdir(pkg)):
        if not f.endswith((".hip", ".h")):
            continue
        for ln, line in enumerate(open(os.path.join(pkg, f), errors="ignore").read().split("\n"), 1):
            if "asm" in line and re.search(r"(global|flat|buffer|scratch)_store_(dwordx[34]|b96|b128)", line):
                seen += 1
                assert re.search(r"store_\w+[^\"]*\\n\\ts_nop [1-9]", line), "%s:%d: wide inline-assembly store without s_nop behind it" % (f, ln)
    assert seen >= 2      # st_pack_wt (pipe.hip), pl_store_wt (lanczos_pl.hip)


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


@pytest.mark.parametrize("hessenberg", [False, True])
def test_host_expm_fp64_block_remainders(eu, hessenberg):
    """The fp64 products and triangular solves of host_dense.h run on 8-row x 6-column register blocks (masked rows,
    1..5 remainder columns, zero tails of Hessenberg factors skipped): every size class against the oracle."""
    rng = np.random.default_rng(11)
    for n in [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 23, 24, 25, 30, 31, 32, 33, 47, 48, 49, 65]:
        A = rng.standard_normal((n, n))
        if hessenberg:
            A = np.triu(A, -1)
        A *= 4.5 / max(np.linalg.norm(A, 1), 1e-300)          # Pade 13, no squaring: the longest chain of products
        E = eu.host_expm(A)
        R = ko.exponential_(A)
        assert np.linalg.norm(E - R) / np.linalg.norm(R) < 1e-13, n
        A *= 20.0                                              # ... and with squarings
        E = eu.host_expm(A)
        R = ko.exponential_(A)
        assert np.linalg.norm(E - R) / np.linalg.norm(R) < 5e-12, n


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


@pytest.mark.parametrize("t", [0.7, -2.5, 0.3 - 1.1j, -1j])
@pytest.mark.parametrize("n", [1, 2, 5, 30, 97])
def test_host_symtridiag_exp_last_is_the_last_entry_bit_for_bit(eu, t, n):
    """the per-step stopping test of the error-estimate mode reads e_j' exp(t T_j) e_1 only; it is computed from
    the first and last eigenvector rows (O(j^2) per step) and must be EXACTLY the entry the full product gives, so that the
    stopping step cannot move -- including on a Lanczos matrix with ghost eigenvalue clusters."""
    rng = np.random.default_rng(n)
    d = rng.standard_normal(n)
    e = np.abs(rng.standard_normal(max(n - 1, 0))) + 0.1
    if n > 10:                                  # clustered spectrum: several nearly equal diagonal entries, tiny couplings
        d[3:8] = 5.0 + 1e-9 * rng.standard_normal(5)
        e[3:7] = 1e-7
    full = eu.host_symtridiag_expcol(d, e, t)
    last = eu.host_symtridiag_exp_last(d, e, t)
    assert last.real == full[-1].real and last.imag == full[-1].imag
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    truth = sl.expm(t * T)[-1, 0]
    assert abs(last - truth) <= 1e-12 * max(1.0, abs(sl.expm(t * T)).max())


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
    assert i["sell"] and 1 <= i["sell_cut"] <= 4 and "overflow" in i["path"]               # one 400-entry row: slots up to a small cut + overflow
    assert i["pipeline_dia_diagonals"] == 0 and i["general_dia_diagonals"] == 0 and i["sell_wave_reach"] == -1
    assert eu.host_pattern_info(c2)["sell_cut"] == 0 and eu.host_pattern_info(irr)["sell_cut"] == 0
    # heavy padding but no useful cut (every slice: half its rows empty, the others all alike) is a REGULAR-row
    # pattern -- no overflow, the diagonal forms stay available
    half = sp.diags([np.where(np.arange(n) % 2 == 0, 1.0, 0.0)], [0], shape=(n, n), format="csr")
    half.eliminate_zeros()
    i = eu.host_pattern_info(half)
    assert i["sell"] and i["sell_cut"] == 0 and "overflow" not in i["path"]
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
    # in-place PERMUTATIONS of the values leave a plain sum unchanged -- the checksum is order-sensitive
    A = sp.diags([np.arange(1.0, 2000.0), np.arange(5.0, 2005.0), np.arange(2.0, 2001.0)], [-1, 0, 1], format="csr")
    f0 = api._fingerprint(A)
    A.data[:] = A.data[::-1].copy()
    assert api._fingerprint(A) != f0
    A.data[:] = A.data[::-1].copy()
    assert api._fingerprint(A) == f0
    A.data[[10, 20]] = A.data[[20, 10]]                      # two entries trade places
    assert api._fingerprint(A) != f0
    S = np.arange(16.0).reshape(4, 4)
    S = S + S.T
    S[0, 3] += 1.0
    g0 = api._fingerprint(S)
    S[:] = S.T.copy()                                         # in-place transpose of the values
    assert api._fingerprint(S) != g0
    F = np.asfortranarray(np.arange(12.0).reshape(3, 4))
    h0 = api._fingerprint(F)                                  # F-ordered: checksummed through its transpose view, no copy
    F[1, 2], F[2, 1] = F[2, 1], F[1, 2]
    assert api._fingerprint(F) != h0


def test_struct_layouts_match_the_library(eu):
    """Every options / result struct a host language restates (ctypes here, `struct` in julia/MIKrylov.jl) against what
    the library was compiled with: size, field order, offsets and types (expv_mi_abi_sizeof / expv_mi_abi_layout)."""
    import ctypes as C
    from exponentialutilities_jl_amd import _lib as L
    lib = L.load()
    tyname = {C.c_int32: "i32", C.c_int64: "i64", C.c_double: "f64", C.c_void_p: "ptr", L.PRINT_FN: "ptr"}
    for name, kind in L.ABI_KINDS.items():
        st = getattr(L, name)
        assert lib.expv_mi_abi_sizeof(kind) == C.sizeof(st), name
        mine = ",".join("%s:%s@%d" % (f, tyname[t], getattr(st, f).offset) for f, t in st._fields_)
        assert lib.expv_mi_abi_layout(kind).decode() == mine, (name, lib.expv_mi_abi_layout(kind).decode(), mine)
    assert lib.expv_mi_abi_sizeof(99) == 0 and lib.expv_mi_abi_layout(99) == b""
    # the Julia shim restates the same structs: its field lists must be the library's, in order
    src = open(os.path.join(ROOT, "julia", "MIKrylov.jl")).read()
    for jl, kind in (("ArnoldiOpts", 0), ("ExpvStats", 1), ("TimestepOpts", 2), ("TimestepStats", 3), ("KiopsOpts", 4)):
        body = re.search(r"struct %s\n(.*?)\nend" % jl, src, flags=re.S).group(1)
        fields = [l.split("::")[0].strip() for l in body.splitlines() if "::" in l]
        want = [f.split(":")[0] for f in lib.expv_mi_abi_layout(kind).decode().split(",")]
        assert fields == want, (jl, fields, want)
        jt = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cdouble": "f64", "Ptr{Cvoid}": "ptr"}
        types = [jt[l.split("::")[1].split("#")[0].strip()] for l in body.splitlines() if "::" in l]
        assert types == [f.split(":")[1].split("@")[0] for f in lib.expv_mi_abi_layout(kind).decode().split(",")], jl


def _split_top(argstr):
    out, depth, cur = [], 0, ""
    for ch in argstr:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_julia_shim_calls_match_the_header():
    """julia/MIKrylov.jl cannot run here (no Julia in the image): at least every ccall in it must name a symbol the header
    declares, with the header's number of arguments, pointer arguments as pointers and scalars as scalars."""
    hdr = open(os.path.join(ROOT, "include", "expv_mi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|size_t|const char \*)\s*\*?\s*(expv_mi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = [] if args in ("", "void") else _split_top(args)
    src = open(os.path.join(ROOT, "julia", "MIKrylov.jl")).read()
    calls = 0
    for m in re.finditer(r"ccall\(\(:(expv_mi_[a-z0-9_]+), lib\),\s*([A-Za-z{}]+),\s*\(", src):
        name = m.group(1)
        assert name in protos, "MIKrylov.jl calls %s, which include/expv_mi.h does not declare" % name
        i, depth = m.end(), 1                     # the Julia argument-type tuple: balanced parentheses from here
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        jl = [a for a in _split_top(src[m.end(): i - 1]) if a]
        c = protos[name]
        assert len(jl) == len(c), (name, jl, c)
        for ja, ca in zip(jl, c):
            c_is_ptr = "*" in ca or "[" in ca or "_fn" in ca or ca.split()[0].endswith("_t") and "expv_mi_" in ca and "int" not in ca.split()[0]
            j_is_ptr = ja.startswith(("Ptr{", "Ref{", "Cstring"))
            assert c_is_ptr == j_is_ptr, (name, ja, ca)
            if not c_is_ptr:
                width = {"int": "Cint", "int32_t": "Cint", "int64_t": "Int64", "double": "Cdouble", "size_t": "Csize_t"}
                assert ja == width[ca.replace("const", "").split()[0]], (name, ja, ca)
        calls += 1
    assert calls >= 25
    used = set(re.findall(r":(expv_mi_[a-z0-9_]+), lib", src))
    for must in ("expv_mi_arnoldi", "expv_mi_lanczos", "expv_mi_expv_ks", "expv_mi_phiv_ks", "expv_mi_phiv_timestep", "expv_mi_kiops",
                 "expv_mi_timestep_caches_create", "expv_mi_expv_error_estimate", "expv_mi_expv", "expv_mi_expv_batch_multi",
                 "expv_mi_abi_sizeof", "expv_mi_ks_resize", "expv_mi_op_create_callback", "expv_mi_ctx_set_option",
                 "expv_mi_ctx_get_option", "expv_mi_ctx_counters"):
        assert must in used, must


@pytest.mark.parametrize("T,tol", [(np.float64, 1e-11), (np.float32, 1e-4), (np.complex128, 1e-11), (np.complex64, 1e-4)])
@pytest.mark.parametrize("scale", [3.0, 1.5, 0.5, 0.1, 0.005])
def test_host_expm_across_blasfloat_types(eu, T, tol, scale):
    """test/basictests.jl:952-974: exponential!(_, ExpMethodHigham2005Base) for every BlasFloat across every Pade norm
    range (C13 with scaling-squaring, C9, C7, C5, C3), against a high-precision reference; the result keeps the type."""
    rng = np.random.default_rng(7)
    n = 40
    A0 = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if np.dtype(T).kind == "c" else 0)
    A = (scale * A0 / np.linalg.norm(A0, 1)).astype(T)
    ref = sl.expm(A.astype(np.complex128))
    E = eu.host_expm(A)
    assert E.dtype == np.dtype(T)
    assert np.linalg.norm(E.astype(np.complex128) - ref) / np.linalg.norm(ref) < tol
    if scale == 3.0:     # phiv_dense through the same routine, same element type
        v = A[:, 0].copy()
        w = eu.host_phiv_dense(A, v, 2)
        assert w.dtype == np.dtype(T)
        w64 = eu.host_phiv_dense(A.astype(np.complex128), v.astype(np.complex128), 2)
        assert np.linalg.norm(w.astype(np.complex128) - w64) / np.linalg.norm(w64) < 10 * tol


def _np_content_hash(a):
    """The definition of expv_mi_host_wrapsum restated in numpy (test infrastructure): sum_i mix64(x_i ^ (i + 1) g) mod 2^64."""
    M = 0xFFFFFFFFFFFFFFFF
    a = np.ascontiguousarray(a).view(np.uint8).ravel()
    nbytes = a.size
    k = nbytes // 8 * 8
    g = np.uint64(0x9e3779b97f4a7c15)

    def mix(h):
        h = h ^ (h >> np.uint64(33)); h = h * np.uint64(0xff51afd7ed558ccd)
        h = h ^ (h >> np.uint64(33)); h = h * np.uint64(0xc4ceb9fe1a85ec53)
        return h ^ (h >> np.uint64(33))
    total = 0
    with np.errstate(over="ignore"):
        if k:
            words = a[:k].copy().view(np.uint64)
            salt = np.arange(1, words.size + 1, dtype=np.uint64) * g
            total = int(np.add.reduce(mix(words ^ salt), dtype=np.uint64))
        if nbytes > k:
            x = int.from_bytes(bytes(a[k:]), "little")
            salt = ((k // 8 + 1) * 0x9e3779b97f4a7c15) & M
            total += int(mix(np.array([x ^ salt ^ ((nbytes - k) << 56)], dtype=np.uint64))[0])
    return (k // 8, total & M)


def test_library_wrapsum_equals_the_numpy_definition(eu):
    """expv_mi_host_wrapsum (threaded, in the library) against its definition restated in numpy: sizes around the thread-split
    thresholds, unaligned starts, tails of 0..7 bytes."""
    import ctypes as C
    from exponentialutilities_jl_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(5)
    for nbytes in (0, 1, 7, 8, 9, 4096 + 3, (1 << 21) + 5, 3 * (1 << 21) + 8, 40_000_013):
        base = rng.integers(0, 256, size=nbytes + 16, dtype=np.uint8)
        for shift in (0, 3):
            a = base[shift:shift + nbytes]
            out = (C.c_uint64 * 2)()
            assert lib.expv_mi_host_wrapsum(a.ctypes.data if nbytes else None, nbytes, out) == 0
            assert (int(out[0]), int(out[1])) == _np_content_hash(a), (nbytes, shift)


def test_content_hash_sees_permutations_of_mantissa_free_values(eu):
    """the linear index-weighted sum of round 3 collided on exactly the values stencils are made of -- 1.0 and 2.0
    trading places at a distance of 2048 k words, a reversed {-2, 1} stencil, permutations of equal-exponent values.  The
    position-salted mixing hash sees all of them, through the fingerprint the Python mirror uses."""
    import scipy.sparse as sp
    from exponentialutilities_jl_amd import api
    a = np.ones(5000)
    a[2048] = 2.0
    b = a.copy()
    b[0], b[2048] = b[2048], b[0]
    assert api._wrapsum(a) != api._wrapsum(b)                 # (round 3: both (5000, 8939645260330434560))
    for k in (1, 2, 3, 7, 16):
        big = np.ones(2048 * k + 10)
        big[5] = 2.0
        sw = big.copy()
        sw[5], sw[5 + 2048 * k] = sw[5 + 2048 * k], sw[5]
        assert api._wrapsum(big) != api._wrapsum(sw), k
    n = 300_000                                               # large enough for the threaded path (> 2 MB of values)
    A = sp.diags([1.0, -2.0, 1.0], [-2, 0, 1], shape=(n, n), format="csr")      # (the symmetric 1, -2, 1 is its own reverse)
    f0 = api._fingerprint(A)
    assert not np.array_equal(A.data, A.data[::-1])
    A.data[:] = A.data[::-1].copy()                           # reversed in place: only 1.0 / -2.0 trade places
    assert api._fingerprint(A) != f0
    A.data[:] = A.data[::-1].copy()
    assert api._fingerprint(A) == f0
    rng = np.random.default_rng(11)
    vals = np.ldexp(1.0 + rng.integers(0, 8, size=70000) / 8.0, 3)      # equal exponent, three mantissa bits
    h0 = api._wrapsum(vals)
    seen = {h0}
    for _ in range(20):
        i, j = rng.integers(0, vals.size, size=2)
        if vals[i] == vals[j]:
            continue
        w = vals.copy()
        w[i], w[j] = w[j], w[i]
        h = api._wrapsum(w)
        assert h not in seen
        seen.add(h)
    assert api._wrapsum(np.zeros(100)) != api._wrapsum(np.zeros(101))
    assert api._wrapsum(np.zeros(3, dtype=np.uint8)) != api._wrapsum(np.zeros(5, dtype=np.uint8))


def test_free_never_dereferences_the_context_handle(eu):
    """expv_mi_free(ctx, p): the finalizer of a host-language array may run after its context's (Julia at exit, a Python cycle):
    the context handle is not read -- a dangling (here: nonsense) handle with nothing to free returns OK instead of crashing."""
    import ctypes as C
    lib = eu._lib.load() if hasattr(eu, "_lib") else None
    if lib is None:
        import sys
        lib = sys.modules[eu.__name__ + "._lib"].load()
    assert lib.expv_mi_free(C.c_void_p(0x10), C.c_void_p(None)) == 0
    assert lib.expv_mi_free(C.c_void_p(None), C.c_void_p(None)) == 0
    # and the Julia shim's array finalizer neither creates a context nor needs a live one
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "MIKrylov.jl")).read()
    line = [ln for ln in src.splitlines() if ":expv_mi_free" in ln and "finalizer" in ln]
    assert len(line) == 1 and "ctx()" not in line[0] and "isassigned(CTX)" in line[0]


def test_julia_shim_expv_methods_do_not_collide_with_the_reference():
    """The reference defines expv!(w::AbstractVector, t::Real, Ks) and expv!(w::AbstractVector{<:Complex}, t::Complex, Ks)
    (krylov_phiv.jl:200-203, :252-255).  A shim method expv!(w::MIVector, t::Number, Ks::MIKs) is more specific in w and Ks and
    less specific in t: Julia reports that as ambiguous at the first call.  The shim must keep the reference's split, and
    expv(t, A, b; mode = :error_estimate) must not reach the reference's _expv_ee (it builds a host KrylovSubspace)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "MIKrylov.jl")).read()
    three_arg = [ln for ln in src.splitlines() if re.match(r"\s*(function\s+)?expv!\(w::MIVector\{", ln) and "Ks::MIKs" in ln and "A::" not in ln]
    kinds = sorted(re.search(r"t::(\w+)", ln).group(1) for ln in three_arg)
    assert kinds == ["Complex", "Real"], three_arg
    assert "ExponentialUtilities._expv_ee(t::Tt, A::MIOperator{T}, b::MIVector{T}" in src


def _shuffled(A, seed):
    q = np.random.default_rng(seed).permutation(A.shape[0])
    return A[q][:, q].tocsr(), q


def test_host_rcm_recovers_banded_and_grid_orderings(eu):
    """The ordering operator creation computes for unstructured patterns (reorder.h: reverse Cuthill-McKee on A + A', VERDICT r3
    item 1): a permutation; a shuffled 5-diagonal matrix gets its bandwidth back (halo form of the single-pass step), a shuffled
    2-D grid the grid width (wave form), a uniformly random pattern stays where it was; structured and irregular patterns are
    not candidates at all."""
    import scipy.sparse as sp
    n = 30_000
    band = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csr")
    Bs, _ = _shuffled(band, 1)
    perm, info = eu.host_rcm(Bs)
    assert sorted(perm.tolist()) == list(range(n))
    assert info["bandwidth_before"] > n // 2 and info["bandwidth_after"] <= 4
    assert info["form_after"] == "single-pass step, halo form" and info["would_reorder"]
    P = Bs[perm][:, perm].tocsr()                                   # P A P': row i of the result is row perm[i] of A
    assert int(np.max(np.abs(P.tocoo().row - P.tocoo().col))) == info["bandwidth_after"]
    k = 600                                                         # n = 360 000 > 400 tiles: the reach decides
    grid = sp.diags([1.0, 1.0, -4.0, 1.0, 1.0], [-k, -1, 0, 1, k], shape=(k * k, k * k), format="csr")
    Gs, _ = _shuffled(grid, 2)
    perm, info = eu.host_rcm(Gs)
    assert sorted(perm.tolist()) == list(range(k * k))
    assert info["bandwidth_after"] <= 2 * k and info["form_before"] == "two-kernel step"
    assert info["form_after"] == "single-pass step, wave form" and info["would_reorder"]
    assert not eu.host_rcm(grid)[1]["would_reorder"]                # natural ordering of a structured grid: diagonals, left alone
    assert not eu.host_rcm(band)[1]["would_reorder"]
    rng = np.random.default_rng(3)
    m = 300_000
    rows = np.repeat(np.arange(m), 4)
    R = (sp.coo_matrix((np.ones(4 * m), (rows, rng.integers(0, m, size=4 * m))), shape=(m, m)).tocsr() + sp.eye(m)).tocsr()
    perm, info = eu.host_rcm(R)
    assert sorted(perm.tolist()) == list(range(m)) and not info["would_reorder"]        # no ordering helps a random graph
    ragged = sp.lil_matrix((5000, 5000))
    ragged[0, :400] = 1.0
    ragged.setdiag(2.0)
    assert not eu.host_rcm(ragged.tocsr())[1]["would_reorder"]      # irregular rows: an ordering does not change row lengths
    # disconnected pieces, empty rows, a 1 x 1 and an empty matrix
    blocks = sp.block_diag([band[:50, :50], sp.csr_matrix((3, 3)), band[:20, :20]], format="csr")
    perm, _ = eu.host_rcm(_shuffled(blocks, 4)[0])
    assert sorted(perm.tolist()) == list(range(73))
    assert eu.host_rcm(sp.csr_matrix(np.ones((1, 1))))[0].tolist() == [0]
    assert eu.host_rcm(sp.csr_matrix((0, 0)))[0].size == 0
    # an unsymmetric pattern is ordered through A + A'
    U = sp.diags([1.0, 1.0], [0, 3], shape=(400, 400), format="csr")
    perm, info = eu.host_rcm(_shuffled(U, 5)[0])
    assert info["bandwidth_after"] <= 6


def test_host_patch_order_of_2d_grid_stencils(eu):
    """The grid-patch ordering of operator creation (context option patch), host side: recognised patterns
    (5- and 9-point stencils, with and without entries across the row ends, ragged last grid row, Float32 tiles), the ordering is a
    permutation, a tile of 512 (Float32: 1024) consecutive stored rows is a patch of the grid -- every stored tile reads at most a
    ring of ~100 (~130) rows outside itself, counted here from the permuted pattern itself -- and patterns that are not 2-D grids
    (banded, 3-D grid, short grid rows, irregular rows) are left alone."""
    import scipy.sparse as sp

    def rings(A, perm, tr):
        P = A[perm][:, perm].tocsr()
        n = P.shape[0]
        out = []
        for t0 in range(0, n, tr):
            cols = P[t0:t0 + tr].indices
            out.append(np.unique(cols[(cols < t0) | (cols >= t0 + tr)]).size)
        return np.array(out)

    k, rows = 200, 190
    n = k * rows
    pure = sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    perm, cnt, info = eu.host_patch_order(pure)
    assert info["patch_form"] and info["grid_row_length"] == k and info["tiles"] == (n + 511) // 512
    assert sorted(perm.tolist()) == list(range(n))
    r = rings(pure, perm, 512)
    assert np.array_equal(r, cnt) and r.max() == info["longest_ring"] <= 160 and 80 < info["mean_ring"] < 110, (r.max(), info)
    assert info["column_indices_stored"] < pure.nnz // 4           # equal column blocks of slices are stored once (n = 10^6: 0.4 %)
    # Float32: tiles of 1024 rows = 32 x 32 patches
    perm32, cnt32, info32 = eu.host_patch_order(pure, np.float32)
    assert info32["patch_form"] and sorted(perm32.tolist()) == list(range(n))
    r32 = rings(pure, perm32, 1024)
    assert np.array_equal(r32, cnt32) and r32.max() <= 256 and 110 < info32["mean_ring"] < 150, (r32.max(), info32)
    # no entries across the row ends (a kron-structured operator), 9-point stencil, ragged last grid row
    i = np.arange(n)
    keep = lambda off: ((i + off >= 0) & (i + off < n) & (((i + off) // k == i // k) if abs(off) == 1 else True))
    lap = sum(sp.csr_matrix((np.ones(keep(o).sum()), (i[keep(o)], i[keep(o)] + o)), shape=(n, n)) for o in (-k, -1, 0, 1, k)).tocsr()
    perm, cnt, info = eu.host_patch_order(lap)
    assert info["patch_form"] and sorted(perm.tolist()) == list(range(n)) and np.array_equal(rings(lap, perm, 512), cnt) and cnt.max() <= 128
    nine = sp.diags([1.0] * 9, [-k - 1, -k, -k + 1, -1, 0, 1, k - 1, k, k + 1], shape=(n, n), format="csr")
    perm, cnt, info = eu.host_patch_order(nine)
    assert info["patch_form"] and np.array_equal(rings(nine, perm, 512), cnt) and cnt.max() <= 200, cnt.max()
    nr = k * 77 + 31
    ragged = sp.diags([1.0] * 5, [-k, -1, 0, 1, k], shape=(nr, nr), format="csr")
    perm, cnt, info = eu.host_patch_order(ragged)
    assert info["patch_form"] and sorted(perm.tolist()) == list(range(nr)) and np.array_equal(rings(ragged, perm, 512), cnt)
    # not 2-D grids
    band = sp.diags([1.0] * 5, [-2, -1, 0, 1, 2], shape=(n, n), format="csr")
    assert eu.host_patch_order(band)[0] is None
    k3 = 30
    g3 = sp.diags([1.0] * 7, [-k3 * k3, -k3, -1, 0, 1, k3, k3 * k3], shape=(k3 ** 3, k3 ** 3), format="csr")
    assert eu.host_patch_order(g3)[0] is None
    short = sp.diags([1.0] * 5, [-40, -1, 0, 1, 40], shape=(40 * 900, 40 * 900), format="csr")      # grid rows shorter than 64 cells
    assert eu.host_patch_order(short)[0] is None
    permc, cntc, infoc = eu.host_patch_order(pure, np.complex128)      # ComplexF64: tiles of 256 rows = 16 x 16 patches
    assert infoc["patch_form"] and infoc["tiles"] == (n + 255) // 256 and sorted(permc.tolist()) == list(range(n))
    rc = rings(pure, permc, 256)
    assert np.array_equal(rc, cntc) and rc.max() <= 128 and 55 < infoc["mean_ring"] < 80, (rc.max(), infoc)
    assert eu.host_patch_order(sp.csr_matrix((0, 0)))[0] is None


def test_host_mesh_patches_of_meshes_in_any_numbering(eu):
    """reorder.h: mesh_patches (context option patch; VERDICT r3 item 1 taken further): a planar mesh numbered at random is cut into
    tiles that are compact blobs of the graph -- the rings, counted here from the permuted pattern itself, stay near 80-100 rows for
    512-row tiles on a square, an L-shaped and a two-component domain and on a triangulation; the ordering is a permutation; a random
    graph (levels of n / 4 nodes) and a 3-D grid are given up."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)

    def planar(k, rows, tri=False, mask=None):
        n = k * rows
        i = np.arange(n)
        parts = []
        for dr, dc in [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)] + ([(1, 1), (-1, -1)] if tri else []):
            r, c = i // k + dr, i % k + dc
            ok = (r >= 0) & (r < rows) & (c >= 0) & (c < k)
            parts.append(sp.csr_matrix((np.ones(ok.sum()), (i[ok], (r * k + c)[ok])), shape=(n, n)))
        A = sum(parts).tocsr()
        if mask is not None:
            keep = np.nonzero(mask(i // k, i % k))[0]
            A = A[keep][:, keep].tocsr()
        q = rng.permutation(A.shape[0])
        return A[q][:, q].tocsr()

    def rings(A, perm, tr=512):
        P = A[perm][:, perm].tocsr()
        out = []
        for t0 in range(0, P.shape[0], tr):
            cols = P[t0:t0 + tr].indices
            out.append(np.unique(cols[(cols < t0) | (cols >= t0 + tr)]).size)
        return np.array(out)

    cases = {"square": planar(300, 280), "L": planar(320, 320, mask=lambda r, c: ~((r > 150) & (c > 170))),
             "triangulation": planar(260, 250, tri=True),
             "two components + isolated": sp.block_diag([planar(200, 150), planar(90, 140, tri=True), sp.csr_matrix((500, 500))], format="csr")}
    for name, A in cases.items():
        perm, cnt, info = eu.host_patch_order(A, mesh=True)
        assert perm is not None and info["patch_form"], name
        assert sorted(perm.tolist()) == list(range(A.shape[0])), name
        r = rings(A, perm)
        assert np.array_equal(r, cnt) and r.max() <= 256 and r.mean() < 115, (name, r.max(), r.mean())
    m = 100_000
    rows = np.repeat(np.arange(m), 4)
    R = (sp.coo_matrix((np.ones(4 * m), (rows, rng.integers(0, m, size=4 * m))), shape=(m, m)).tocsr() + sp.eye(m)).tocsr()
    assert eu.host_patch_order(R, mesh=True)[0] is None
    k3 = 40
    g3 = sp.diags([1.0] * 7, [-k3 * k3, -k3, -1, 0, 1, k3, k3 * k3], shape=(k3 ** 3, k3 ** 3), format="csr")
    assert eu.host_patch_order(g3, mesh=True)[0] is None                # levels of a 3-D grid are planes: far wider than 8 sqrt(n)
    assert eu.host_patch_order(sp.identity(100, format="csr"), mesh=True)[0] is None      # (too small to bother)


def test_host_ordering_entry_points_refuse_malformed_patterns(eu):
    """expv_mi_host_rcm / _host_patch_order / _host_mesh_patch_order index with the caller's rowptr / colind; a column
    index outside [0, n), a decreasing rowptr or rowptr[0] != 0 is an ArgumentError like in expv_mi_op_create_csr, not a heap
    overrun."""
    from exponentialutilities_jl_amd import _lib
    lib = _lib.load()
    n = 6
    good_rp = np.array([0, 2, 4, 6, 8, 10, 12], dtype=np.int32)
    good_ci = np.array([0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0], dtype=np.int32)
    perm = np.zeros(n, dtype=np.int32)
    cnt = np.zeros(4, dtype=np.int32)
    out4, out8 = np.zeros(4, dtype=np.int64), np.zeros(8, dtype=np.int64)
    F64, ARG = 0, 2      # EXPV_MI_F64, EXPV_MI_ARGUMENT_ERROR (include/expv_mi.h)
    assert lib.expv_mi_host_rcm(n, good_rp.ctypes.data, good_ci.ctypes.data, F64, perm.ctypes.data, out4.ctypes.data) == 0
    assert sorted(perm.tolist()) == list(range(n))
    bad = []
    ci = good_ci.copy(); ci[5] = n          # column == n
    bad.append((good_rp, ci))
    ci = good_ci.copy(); ci[0] = -1         # negative column
    bad.append((good_rp, ci))
    rp = good_rp.copy(); rp[3] = 3          # decreasing rowptr
    bad.append((rp, good_ci))
    rp = good_rp.copy(); rp[0] = 1          # 1-based rowptr
    bad.append((rp, good_ci))
    for rp, ci in bad:
        assert lib.expv_mi_host_rcm(n, rp.ctypes.data, ci.ctypes.data, F64, perm.ctypes.data, out4.ctypes.data) == ARG
        assert lib.expv_mi_host_patch_order(n, rp.ctypes.data, ci.ctypes.data, F64, perm.ctypes.data, cnt.ctypes.data, out8.ctypes.data) == ARG
        assert lib.expv_mi_host_mesh_patch_order(n, rp.ctypes.data, ci.ctypes.data, F64, perm.ctypes.data, cnt.ctypes.data, out8.ctypes.data) == ARG


def test_orderings_do_not_depend_on_the_locality_relabelling(eu):
    """Round 5: the breadth-first searches of operator creation (reverse Cuthill-McKee, mesh patches) run on a copy of the graph
    renumbered in breadth-first order (reorder.h: Graph::local -- cache-local arrays instead of a miss per node); visiting orders and
    ties follow the ORIGINAL numbering, so every ordering must come out bit for bit as without the copy (EXPV_MI_NO_RELABEL=1):
    shuffled band, shuffled 2-D grid, a shuffled triangulated mesh, two components + isolated nodes, a random graph (given up)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(77)

    def shuffled(A, seed):
        q = np.random.default_rng(seed).permutation(A.shape[0])
        return A[q][:, q].tocsr()
    n = 150_000
    band = shuffled(sp.diags([0.3, 1.2, -2.0, 0.8, -0.1], [-2, -1, 0, 1, 2], shape=(n, n), format="csr"), 1)
    k = 400
    grid0 = sp.diags([1.0, 1.0, -4.0, 1.0, 1.0], [-k, -1, 0, 1, k], shape=(k * k, k * k), format="csr")
    grid = shuffled(grid0, 2)
    ii = np.arange(k * k)
    parts = []
    for dr, dc in ((0, 0), (0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, -1)):
        r, c = ii // k + dr, ii % k + dc
        ok = (r >= 0) & (r < k) & (c >= 0) & (c < k)
        parts.append(sp.csr_matrix((np.ones(ok.sum()), (ii[ok], (r * k + c)[ok])), shape=(k * k, k * k)))
    mesh = shuffled(sum(parts).tocsr(), 3)
    two = shuffled(sp.block_diag([grid0[:80_000, :80_000], sp.identity(7, format="csr"), grid0[:70_000, :70_000]], format="csr"), 4)
    nr = 140_000
    rnd = (sp.csr_matrix((np.ones(4 * nr), (rng.integers(0, nr, 4 * nr), rng.integers(0, nr, 4 * nr))), shape=(nr, nr)) + sp.identity(nr, format="csr")).tocsr()
    rnd.sum_duplicates()
    for name, A in (("band", band), ("grid", grid), ("mesh", mesh), ("two components", two), ("random", rnd)):
        A.sort_indices()
        res = {}
        for mode in ("relabelled", "plain"):
            if mode == "plain":
                os.environ["EXPV_MI_NO_RELABEL"] = "1"
            else:
                os.environ.pop("EXPV_MI_NO_RELABEL", None)
            try:
                perm, info = eu.host_rcm(A)
                pperm, cnt, pinfo = eu.host_patch_order(A, mesh=True)
            finally:
                os.environ.pop("EXPV_MI_NO_RELABEL", None)
            res[mode] = (perm.copy(), info, None if pperm is None else pperm.copy(), cnt.copy(), pinfo)
        a, b = res["relabelled"], res["plain"]
        assert sorted(a[0].tolist()) == list(range(A.shape[0])), name
        assert np.array_equal(a[0], b[0]), "%s: reverse Cuthill-McKee differs with the local copy" % name
        assert a[1] == b[1], name
        assert (a[2] is None) == (b[2] is None), name
        if a[2] is not None:
            assert np.array_equal(a[2], b[2]), "%s: mesh patches differ with the local copy" % name
            assert np.array_equal(a[3], b[3]) and a[4] == b[4], name

"""expv on general (non-banded) sparse operators: which step form runs, per-kernel time, fraction of the SURVEY 8d contract.
    python tools/general_sparse.py [kind ...]     kinds: c2 rand5 band5[_REACH] grid grid3 powerlaw c2f32 mesh trimesh diskmesh, shuf_<kind>; options name=value (context options)"""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
import expv_mi_loader
import bench

eu = expv_mi_loader.load()


def make(kind, n, seed=11):
    rng = np.random.default_rng(seed)
    if kind == "c2":
        return bench.c2_operator(n)
    if kind == "rand5" or kind.startswith("band5"):          # regular rows: diagonal + 4 random columns (anywhere / within +-reach)
        k = 4
        rows = np.repeat(np.arange(n), k)
        if kind == "rand5":
            cols = rng.integers(0, n, size=n * k)
        else:
            reach = int(kind[6:]) if len(kind) > 5 else 20000            # band5_2000: reach 2000
            cols = np.clip(rows + rng.integers(-reach, reach + 1, size=n * k), 0, n - 1)
        vals = rng.standard_normal(n * k) * 0.3
        A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr() + sp.diags([-0.5 * np.ones(n)], [0], format="csr")
        A.sum_duplicates()
        return A.tocsr()
    if kind == "c2f32":                     # the headline operator in Float32 (native 32-bit storage)
        return bench.c2_operator(n).astype(np.float32)
    if kind == "grid":                      # 2-D 5-point stencil, variable coefficients: general DIA form
        k = int(round(np.sqrt(n)))
        d = [0.3 + 0.05 * rng.random(n - k), 1.2 + 0.05 * rng.random(n - 1), -2.0 + 0.05 * rng.random(n), 0.8 + 0.05 * rng.random(n - 1),
             -0.1 + 0.05 * rng.random(n - k)]
        return sp.diags(d, [-k, -1, 0, 1, k], shape=(n, n), format="csr")
    if kind == "gridf32":                   # the 2-D stencil in Float32: wave form on 1024-row tiles
        return make("grid", n, seed).astype(np.float32)
    if kind == "grid3":                     # 3-D 7-point stencil on a k x k x k grid, variable coefficients: general DIA form
        k = int(round(n ** (1.0 / 3.0)))
        nn = k * k * k
        offs = [-k * k, -k, -1, 0, 1, k, k * k]
        base = [0.2, 0.3, 1.1, -2.0, 0.7, -0.1, 0.15]
        d = [c + 0.05 * rng.random(nn - abs(o)) for c, o in zip(base, offs)]
        A = sp.diags(d, offs, shape=(nn, nn), format="csr")
        return sp.block_diag([A, -0.5 * sp.identity(n - nn, format="csr")], format="csr") if n > nn else A
    if kind == "powerlaw":                  # irregular rows: lengths ~ Zipf, mean ~5, max capped
        ln = np.minimum(rng.zipf(1.8, size=n), 2000)
        ln = np.maximum(1, (ln * (5.0 / ln.mean())).astype(np.int64))
        ln = np.minimum(ln, 4000)
        rows = np.repeat(np.arange(n), ln)
        cols = rng.integers(0, n, size=rows.size)
        vals = rng.standard_normal(rows.size) / np.sqrt(np.repeat(ln, ln))
        A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr() + sp.diags([-0.5 * np.ones(n)], [0], format="csr")
        A.sum_duplicates()
        return A.tocsr()
    if kind in ("mesh", "trimesh", "diskmesh"):      # planar meshes (no entries across the row ends of the underlying grid): 5-point, 7-point
        k = int(round(np.sqrt(n)))                   # (a triangulated grid), and a 5-point mesh on a disk; variable coefficients
        i = np.arange(k * k)
        offs = [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)] + ([(1, 1), (-1, -1)] if kind == "trimesh" else [])
        parts = []
        for dr, dc in offs:
            r, c = i // k + dr, i % k + dc
            ok = (r >= 0) & (r < k) & (c >= 0) & (c < k)
            v = (-2.0 if (dr, dc) == (0, 0) else 0.5) + 0.05 * rng.random(k * k)
            parts.append(sp.csr_matrix((v[ok], (i[ok], (r * k + c)[ok])), shape=(k * k, k * k)))
        A = sum(parts).tocsr()
        if kind == "diskmesh":
            keep = np.nonzero((i // k - k / 2) ** 2 + (i % k - k / 2) ** 2 <= (0.49 * k) ** 2)[0]
            A = A[keep][:, keep].tocsr()
        nn = A.shape[0]
        return sp.block_diag([A, -0.5 * sp.identity(n - nn, format="csr")], format="csr") if n > nn else A
    if kind.startswith("shuf_"):            # any of the above under a random symmetric permutation of the unknowns
        A = make(kind[5:], n, seed).tocsr()
        q = np.random.default_rng(seed + 1).permutation(A.shape[0])
        return A[q][:, q].tocsr()
    raise SystemExit("unknown kind " + kind)


def main():
    opts = [a for a in sys.argv[1:] if "=" in a]
    kinds = [a for a in sys.argv[1:] if "=" not in a] or ["c2", "band5", "rand5", "powerlaw"]
    n, m = 1_000_000, 30
    ctx = eu.Context(async_outputs=True)
    for o in opts:
        ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
    b64 = torch.as_tensor(np.random.default_rng(3).standard_normal(n), device="cuda")
    w64 = torch.empty(n, dtype=torch.float64, device="cuda")
    for kind in kinds:
        A = make(kind, n)
        b = b64.to(torch.float32) if A.dtype == np.float32 else b64
        w = torch.empty(n, dtype=torch.float32, device="cuda") if A.dtype == np.float32 else w64
        if kind == "c2":
            ctx.set_option("pipeline", 0)
        t0 = time.perf_counter()
        op = eu.MIOperator(A, ctx)
        t_setup = time.perf_counter() - t0
        info = eu.host_pattern_info(A, A.dtype)
        f = lambda: eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
        f(); ctx.sync()
        t = bench.timed(f, 10, 2, ctx.sync)
        path = eu.expv.last_stats["path"]
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(5):
            f()
        ctx.sync()
        prof = ctx.prof_get(); ctx.prof_enable(False)
        balg = bench.alg_bytes_expv(n, A.nnz, m, s=A.dtype.itemsize)
        rl = np.diff(A.indptr)
        print(json.dumps({"kind": kind, "opts": opts, "n": n, "nnz": int(A.nnz), "row_len_max": int(rl.max()), "row_len_mean": float(rl.mean()),
                          "setup_s": t_setup, "reorder": op.reorder_info, "pattern": {k: (v if isinstance(v, (bool, str)) else int(v)) for k, v in info.items()},
                          "path": list(path), "ms_per_expv": 1e3 * t, "matvecs_per_s": m / t, "alg_GB": balg / 1e9,
                          "frac": balg / t / 8e12,
                          "kernels": {k: {"n": v["launches"] // 5, "avg_us": 1e3 * v["total_ms"] / v["launches"]} for k, v in prof.items()}}))
        ctx.set_option("pipeline", 1)
        del op


if __name__ == "__main__":
    main()

"""CPU stand-in with the product's Python signatures, for ``python bench.py --gpus N --standin tests.standin_eu``.

TEST INFRASTRUCTURE ONLY.  bench.py's launch, sharding, collective and verification code has to be exercised at N > 1 on a box
without GPUs; this module supplies the solver behind the same call signatures (MIOperator / expv / expv_batch / phiv_timestep),
backed by the oracle (the checker of the -m gpu tests), on CPU tensors under gloo.  Nothing here is the product path, nothing
here is measured, and bench.py labels every line produced with it as "standin"."""
import numpy as np
import torch

from oracle import krylov_oracle as ko


class _Op:
    def __init__(self, A=None, ctx=None, dtype=None, ishermitian=None, matvec=None, shape=None):
        self.A, self._mv = A, matvec
        self.shape = tuple(shape) if shape is not None else A.shape
        self.dtype = np.dtype(dtype or np.float64)
        self.ishermitian = bool(ishermitian)

    def __matmul__(self, x):
        if self._mv is not None:
            return self._mv(torch.as_tensor(np.ascontiguousarray(x))).numpy().copy()
        return self.A @ x


def _host(x):
    return x.numpy() if hasattr(x, "numpy") else np.asarray(x)


class _Expv:
    last_stats = {}

    def __call__(self, t, A, b, m=30, ishermitian=False, out=None, **kw):
        M = A.A if isinstance(A, _Op) else A
        w = ko.expv(t, M.tocsr() if hasattr(M, "tocsr") else M, _host(b), m=m, ishermitian=ishermitian)
        _Expv.last_stats = {"m": m, "path": ["standin"]}
        if out is not None:
            out.copy_(torch.as_tensor(w))
            return out
        return w


class StandIn:
    MIOperator = _Op
    expv = _Expv()

    @staticmethod
    def expv_batch(t, A0, vals, B, m=30, ctx=None, **kw):
        vals, Bn = _host(vals), _host(B)
        cols = []
        for p in range(vals.shape[0]):
            Ap = A0.copy()
            Ap.data = vals[p].copy()
            cols.append(ko.expv(t, Ap, Bn[:, p], m=m, ishermitian=False))
        return torch.as_tensor(np.stack(cols, axis=1) if cols else np.zeros((A0.shape[0], 0)))

    @staticmethod
    def phiv_timestep(ts, op, B, adaptive=True, tol=1e-7, m=10, stats=None, **kw):
        st = {}
        U = ko.phiv_timestep(ts, op, np.asfortranarray(_host(B)), adaptive=adaptive, tol=tol, m=m,
                             ishermitian=getattr(op, "ishermitian", False), stats=st)
        if stats is not None:
            stats.update(st)
        return torch.as_tensor(np.asarray(U))

"""Where the time of a dense operator built from a row-major numpy array goes (numpy transposing copy vs the library call vs a plain upload)."""
import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
eu = expv_mi_loader.load()
n = 8192
A = np.random.default_rng(0).standard_normal((n, n)) / np.sqrt(n)
t0 = time.perf_counter(); M = np.asfortranarray(A, dtype=np.float64); print("asfortranarray ms", round(1e3 * (time.perf_counter() - t0), 1))
t0 = time.perf_counter(); op = eu.MIOperator(M); torch.cuda.synchronize(); print("MIOperator(F-ordered) ms", round(1e3 * (time.perf_counter() - t0), 1))
t0 = time.perf_counter(); d = torch.as_tensor(M.T.copy() if False else M, device="cuda"); torch.cuda.synchronize(); print("torch upload ms", round(1e3 * (time.perf_counter() - t0), 1))

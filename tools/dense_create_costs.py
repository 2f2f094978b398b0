"""Creation time of a dense operator from a host array and from a device-resident tensor, and whether both report the same properties."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
eu = expv_mi_loader.load()
for n in (2048, 8192):
    A = np.random.default_rng(0).standard_normal((n, n)) / np.sqrt(n)
    t0 = time.perf_counter(); op = eu.MIOperator(A); torch.cuda.synchronize(); t1 = time.perf_counter()
    Ad = torch.as_tensor(A, device="cuda")
    torch.cuda.synchronize(); t2 = time.perf_counter(); opd = eu.MIOperator(Ad); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(n, "dense host create ms", round(1e3 * (t1 - t0), 1), "| device-resident create ms", round(1e3 * (t3 - t2), 2), "| herm", op.ishermitian, opd.ishermitian, "opnorm equal", op.opnorm_inf == opd.opnorm_inf)

"""the overlapped single-pass step relies on two kernels being co-resident with bounded spin-waits; its redo
path has been stressed with other COMPUTE processes on the device, never with RCCL kernels occupying CUs of the same GPU -- which is
what the final gather of a sharded batch is.  This runs the headline call (expv, n = 1e6, m = 30, overlapped form) in a loop WHILE
a second stream of the same device runs RCCL all-gathers of 102 MB blocks back to back (world size 1 is all a 1-GPU box allows: the
RCCL kernel and its channel workgroups are the same) -- once through torch.distributed (backend nccl) and once through the C ABI's own
gather (expv_mi_gather_rccl) -- and reports: calls, results bitwise equal to the quiet reference, the context's redo counters, the
rate with and without the collective traffic.       python tools/stress_rccl.py [seconds per phase]"""
import os
import sys
import threading
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import torch.distributed as dist
import expv_mi_loader
from bench import c2_operator

eu = expv_mi_loader.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
n, m = 1_000_000, 30
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(c2_operator(n), ctx)
g = torch.Generator(device="cuda"); g.manual_seed(3)
b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
w = torch.empty_like(b)
eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
ctx.sync()
ref = w.clone()


def headline_loop(seconds, label):
    c0 = ctx.counters()
    t0 = time.perf_counter()
    calls = bad = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            eu.expv(1.0, op, b, m=m, ishermitian=False, out=w)
            calls += 1
        ctx.sync()
        if not torch.equal(w, ref):
            bad += 1
    dt = time.perf_counter() - t0
    c1 = ctx.counters()
    redo = {k: c1[k] - c0[k] for k in c1 if k.startswith("redo")}
    print("%-44s %6d calls  %.4f ms per call  results differing from the quiet reference: %d  redo counters: %s  overlapped factorisations: %d"
          % (label, calls, 1e3 * dt / calls, bad, redo, c1["overlapped"] - c0["overlapped"] if "overlapped" in c1 else -1), flush=True)
    return calls, bad, redo


print("counters available:", sorted(ctx.counters()))
headline_loop(budget, "quiet device")

# ---- RCCL through torch.distributed on a second stream ----
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
blk = torch.randn(102_400_000 // 8, dtype=torch.float64, device="cuda")      # 102 MB: one rank's result block of BASELINE configs[4]
out = torch.empty_like(blk)
stop = threading.Event()
gathers = [0]


def torch_traffic():
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        while not stop.is_set():
            for _ in range(4):
                dist.all_gather_into_tensor(out, blk)
                gathers[0] += 1
            s2.synchronize()


th = threading.Thread(target=torch_traffic)
th.start()
time.sleep(0.3)
headline_loop(budget, "under torch.distributed all_gather (nccl)")
stop.set(); th.join()
print("   all-gathers of 102 MB meanwhile: %d (%.1f GB/s of copy traffic)" % (gathers[0], 2 * 0.1024 * gathers[0] / budget))
assert torch.equal(out, blk)
dist.destroy_process_group()

# ---- RCCL through the C ABI on its own context (= its own stream) ----
if eu.rccl_available():
    ctx2 = eu.Context(async_outputs=True)
    comm = eu.RcclComm(ctx2, eu.rccl_unique_id(), 1, 0)
    stop.clear(); gathers[0] = 0

    def abi_traffic():
        while not stop.is_set():
            for _ in range(4):
                comm.all_gather(blk, out=out)
                gathers[0] += 1
            ctx2.sync()

    th = threading.Thread(target=abi_traffic)
    th.start()
    time.sleep(0.3)
    headline_loop(budget, "under expv_mi_gather_rccl (C ABI)")
    stop.set(); th.join()
    print("   all-gathers of 102 MB meanwhile: %d (%.1f GB/s of copy traffic)" % (gathers[0], 2 * 0.1024 * gathers[0] / budget))
    assert torch.equal(out, blk)
    comm.destroy()
else:
    print("librccl.so not available through the C ABI")

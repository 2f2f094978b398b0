"""Streaming-bandwidth probe (HBM vs Infinity Cache) used to choose the kernel structure:
repeated device-to-device copies and reductions over buffers of increasing size."""
import json
import sys
import torch

def timeit(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

res = []
for mb in (8, 32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    reps = max(5, min(200, 8192 // mb))
    t_copy = timeit(lambda: y.copy_(x), reps)
    t_sum = timeit(lambda: x.sum(), reps)
    t_dot = timeit(lambda: torch.dot(x, y), reps)
    res.append({"MB": mb, "copy_GBps": 2 * mb / 1024 / t_copy, "sum_GBps": mb / 1024 / t_sum,
                "dot_GBps": 2 * mb / 1024 / t_dot})
    print(res[-1], flush=True)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/membench.json", "w"), indent=1)

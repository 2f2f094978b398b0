"""A few adaptive expv_timestep calls at n = 1e6 (config-2 operator): the command rocprofv3 traces for the integrator-path timeline."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import expv_mi_loader
from bench import c2_operator
eu = expv_mi_loader.load()
n = 1_000_000
ctx = eu.Context(async_outputs=True)
op = eu.MIOperator(c2_operator(n), ctx)
b = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(4):
    U = eu.expv_timestep([0.5, 1.0], op, b, tol=1e-6, adaptive=True)
ctx.sync()

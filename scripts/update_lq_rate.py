#!/usr/bin/env python3
"""HBM rate of the device-resident LQ assembly kernel (gar_update_lq) at the bench shape."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd.gar import BatchedRiccatiSolver

N, nx, nu = 256, 36, 12
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=batch)
dd = s.deriv_doubles
dev = torch.randn(batch * dd, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    s.update_lq_subproblem_device(dev.data_ptr(), 1e-8, True)
s.sync()
t0 = time.perf_counter(); K = 5
for _ in range(K):
    s.update_lq_subproblem_device(dev.data_ptr(), 1e-8, True)
s.sync()
dt = (time.perf_counter() - t0) / K
byt = 8.0 * batch * (dd + s.problem_doubles)
print(f"gar_update_lq batch {batch}: {dt*1e3:.3f} ms, {byt/dt/1e9:.0f} GB/s (read derivs {dd} + write knots {s.problem_doubles} doubles/problem)")

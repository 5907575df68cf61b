"""The headline workload (N=256, nx=36, nu=12, generator W) over the batch size: sweeps/s, the backward kernel's
time and fraction of the 8 TB/s roofline, which kernel family the library picks."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
bytes_bwd = 8 * ((2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu) + ((nu + nx) * (nx + 1) + nx * nx + nx)) * N
for B in (1, 64, 256, 512, 1024, 2048, 4096, 8192):
    s = BatchedRiccatiSolver(dims, nx, batch=B)
    synth_device.fill_problems(s, seed=1, mode="W")
    for _ in range(2):
        s.backward_async(1e-14); s.forward_async()
    s.sync()
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    reps, kb = 5, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        s.backward_async(1e-14); s.forward_async()
        o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o)); kb += o[0]
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"batch {B:5d}  {s.kernel_name:12s} {B / dt:10.0f} sweeps/s  {dt * 1e3:8.3f} ms/step  backward {kb / reps:7.3f} ms = "
          f"{bytes_bwd * B / (kb / reps * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)
    s.close()

import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, mueq = 36, 12, 256, 1e-12
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
dims = [k.dims for k in prob.stages]
import ctypes as C
for B in (1, 16, 64, 128, 256):
    for fam in ("wg4", "wave", "pair"):
        os.environ["GAR_HIP_BACKWARD"] = fam
        s = BatchedRiccatiSolver(dims, nx, batch=B)
        pk = s.pack(prob)
        for b in range(B): s.upload_packed(pk, b, 1)
        s.backward(mueq); s.forward()
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        best = 1e9; kb = 0
        for _ in range(10):
            t0 = time.perf_counter(); s.backward_async(mueq); s.forward_async(); s.sync(); dt = time.perf_counter() - t0
            if dt < best:
                best = dt
                o = (C.c_double * 3)(); s._L.gar_hip_last_kernel_ms(s.handle, o); kb = o[0]; kf = o[1]
        print(f"batch {B:4d} {s.kernel_name:12s} {best*1e6:9.1f} us / sweep  (backward kernel {kb*1e3:8.1f} us, forward {kf*1e3:8.1f} us)", flush=True)
        s.close()

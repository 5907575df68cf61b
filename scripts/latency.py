"""Secondary figures for DESIGN.md: single-problem sweep latency (serial and leg-parallel on
one GPU) and the PCIe-inclusive rate of the host-buffer boundary (upload_packed + sweep +
get_solution), none of which is the headline metric."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver

nx, nu, N, mueq = 36, 12, 256, 1e-12
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
dims = [k.dims for k in prob.stages]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("== single problem, data resident in HBM: backward + forward latency")
for legs in (1, 4, 8, 16, 32):
    s = BatchedRiccatiSolver(dims if legs == 1 else [(a, b, c, d, 0) for (a, b, c, d, _) in dims], nx, batch=1, num_legs=legs)
    s.upload([prob])
    s.backward(mueq); s.forward()

    def sweep():
        s.backward_async(mueq); s.forward_async(); s.sync()
    print(f"  legs={legs:2d} kernel={s.kernel_name:12s} {timeit(sweep) * 1e6:9.1f} us / sweep")

print("== host-buffer boundary (PCIe inclusive): upload_packed + backward + forward + get_solution")
for B in (1, 64, 1024):
    s = BatchedRiccatiSolver(dims, nx, batch=B)
    packed = np.tile(s.pack(prob), B)
    X = np.zeros(int(s.dims[:, 0].sum())); U = np.zeros(int(s.dims[:, 1].sum()))
    L = np.zeros(nx + N * nx); PD = C.POINTER(C.c_double)

    def roundtrip():
        s.upload_packed(packed, 0, B)
        s.backward_async(mueq); s.forward_async()
        for b in range(B):
            s._L.gar_hip_get_solution(s.handle, b, X.ctypes.data_as(PD), U.ctypes.data_as(PD), None, L.ctypes.data_as(PD))
    dt = timeit(roundtrip, reps=5 if B > 64 else 20)
    print(f"  batch={B:5d}: {dt * 1e3:9.3f} ms / step = {B / dt:10.0f} sweeps/s  ({packed.nbytes / 1e6:.1f} MB H2D per step)")

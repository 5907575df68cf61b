"""Single-problem sweep latency, serial vs parallel-in-time (legs) on ONE GPU, with the
per-kernel split (HIP events of the library: leg sweep + tuples | condensed solve | roll-out)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver

nx, nu, mueq = int(os.environ.get("NX", "36")), int(os.environ.get("NU", "12")), 1e-12
batch = int(os.environ.get("BATCH", "1"))
cases = [(256, (1, 8, 16, 32)), (2048, (1, 32, 64, 128, 256))]
if os.environ.get("CASES"):
    cases = eval(os.environ["CASES"])
for N, legs_list in cases:
    prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
    dims = [k.dims for k in prob.stages]
    ref = None
    for legs in legs_list:
        s = BatchedRiccatiSolver(dims, nx, batch=batch, num_legs=legs)
        if legs > 1 and os.environ.get("REFINE") is not None:
            s._check(s._L.gar_hip_set_refinement(s.handle, 1e-10, int(os.environ["REFINE"])))
        s.upload([prob] * batch)
        s.backward(mueq); s.forward(); s.sync()
        sol = s.solution(0)
        if ref is None:
            ref = sol
        err = max(float(np.abs(a - b).max()) for A, B in zip(sol, ref) for a, b in zip(A, B) if a.size)
        sc = max(1.0, max(float(np.abs(v).max()) for v in ref[3]))
        reps = 10
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            s.backward_async(mueq); s.forward_async()
        s.sync()
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):   # one sweep at a time: launch + sync latency included
            s.backward_async(mueq); s.forward_async(); s.sync()
        dts = (time.perf_counter() - t0) / reps
        k = np.zeros(3)
        try:
            s._check(s._L.gar_hip_set_timing(s.handle, 1))
            for _ in range(5):
                s.backward_async(mueq); s.forward_async()
                o = (C.c_double * 3)()
                s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
                k += np.array(list(o))
            k /= 5
        except RuntimeError:
            k[:] = float("nan")  # the generic kernels carry no per-kernel events
        inf = "-"
        if legs > 1:
            o2 = (C.c_double * 2)()
            s._check(s._L.gar_hip_condensed_info(s.handle, 0, o2))
            inf = f"resid {o2[0]:.1e} after {int(o2[1])} refinement steps"
        print(f"N={N:5d} batch={batch} legs={legs:3d} {s.kernel_name:16s} sweep {dt*1e3:8.3f} ms (synced each: {dts*1e3:6.3f}) | "
              f"bwd {k[0]:7.3f} cond/init {k[1]:7.3f} fwd {k[2]:7.3f} ms | rel.diff vs serial {err/sc:.1e} | {inf}", flush=True)

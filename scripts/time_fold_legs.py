"""Leg-mode sweep of one (36, 12, nc) problem, N = 256: per-phase times from the library's own HIP events
(backward legs | condensed | forward) for nc = 0 and nc = 32 at several leg counts."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 36, 12, int(os.environ.get("N", "256"))
for nc in (0, 32):
    prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
    for legs in (6, 32):
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs)
        s.upload([prob])
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        for _ in range(3):
            s.backward_async(1e-11); s.forward_async()
        s.sync()
        acc = np.zeros(3); t0 = time.perf_counter()
        for _ in range(5):
            s.backward_async(1e-11); s.forward_async()
            o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o)); acc += np.array(list(o))
        wall = (time.perf_counter() - t0) / 5 * 1e3
        res, steps = s.condensed_info(0)
        print(f"   condensed residual {res:.2e} steps {steps} omega {s.condensed_backward_error(0):.2e}")
        print(f"nc={nc:2d} legs={legs:3d} {s.kernel_name:22s} backward {acc[0]/5:.3f} condensed {acc[1]/5:.3f} forward {acc[2]/5:.3f} ms  (wall incl. sync {wall:.3f})", flush=True)
        s.close()

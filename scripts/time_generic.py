"""The any-dimension kernels (GAR_HIP_FORCE_GENERIC=1, no padding) on a few shapes: one problem, serial and in leg
mode, and a constrained coupled (D != 0) shape.  Wall times, synchronised after each sweep."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GAR_HIP_PAD", "0")
os.environ.setdefault("GAR_HIP_FORCE_GENERIC", "1")
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
LIB = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else None   # (another build of the library, for an A/B)
for nx, nu, nc, N, legs_list in ((20, 7, 0, 128, (1, 4, 8)), (36, 12, 32, 128, (1,)), (48, 16, 0, 128, (1, 8))):
    prob = synth.generate_lq_problem(11, np.zeros(nx), N, nx, nu, nc=nc, mode="W")   # (nc > 0: dense C and D)
    for legs in legs_list:
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=LIB)
        s.upload([prob])
        for _ in range(2):
            s.backward_async(1e-8); s.forward_async()
        s.sync()
        tb = tf = 0.0
        for _ in range(5):
            t0 = time.perf_counter(); s.backward_async(1e-8); s.sync()
            t1 = time.perf_counter(); s.forward_async(); s.sync()
            t2 = time.perf_counter()
            tb += t1 - t0; tf += t2 - t1
        print(f"nx={nx} nu={nu} nc={nc} N={N} legs={legs:2d} {s.kernel_name:14s} backward(+condensed) {tb/5*1e3:.3f} "
              f"forward {tf/5*1e3:.3f} ms  (wall, synchronised after each)", flush=True)
        s.close()

"""Cycle stamps of two consecutive elimination steps of the condensed solver (problem 0)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, legs = 36, 12, 256, int(os.environ.get("LEGS", "8"))
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")  # make -C aligator_amd/csrc trace
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=TRACE_LIB)
s.upload([prob]); s.backward(1e-12); s.forward()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-12)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 0, out)
names = ["factor", "W=D^-1 (36 rhs)", "store W + matvec", "load B, D_i", "x_i, U=WB^T | -I: all", "D_i -= B U", "next"]
for st in range(2):
    t = [out[16 * st + k] for k in range(7)]
    print(f"step {st} ({'B = Vxt' if t[5] else 'B = -I'}):", end=" ")
    prev = t[0]
    for k in range(1, 7):
        if t[k]:
            print(f"{names[k-1]} {t[k]-prev}", end=" | "); prev = t[k]
    print(f"total {prev - t[0]} (100 MHz ticks: x24 = shader cycles)")

print("cyclic reduction, first level, survivor 2, wave 0 (shader cycles):")
y = [out[32 + k] for k in range(10)]
nm = ["loads S_i,S_j,C_i", "inverse (factor+W)", "store W, Cl", "U=W C^T, S -= C U", "matvecs", "load C_j", "new coupling -U^T C_j",
      "wait wave 1", "combine + store S_i"]
for k in range(1, 10):
    if y[k] and y[k-1]:
        print(f"  {nm[k-1]:28s} {y[k]-y[k-1]:7d}")
print("  total", y[9] - y[0])

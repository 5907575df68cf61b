"""Phase cycle stamps of one constrained stage (nx=36, nu=12, nc=32) of the wave kernel (debug build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")
nx, nu, nc, N = 36, 12, 32, 64
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, lib_path=TRACE_LIB)
s.upload([prob]); s.backward(1e-8)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-8)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))[:11]
names = ["start", "vplus", "qhat", "S1S2", "export", "KKT assemble+factor", "KKT solve (37 rhs)", "Kb+vec+Z out", "Aff", "Vxx+C^TZ", "store"]
print(s.kernel_name, "stage total", t[10] - t[0])
prev = t[0]
for i in range(1, 11):
    if t[i]:
        print(f"  {names[i]:18s} {t[i]-prev:8d}"); prev = t[i]

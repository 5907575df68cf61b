"""Phase-by-phase cycle breakdown of one parameterised stage of the wave-leg backward kernel."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")  # make -C aligator_amd/csrc trace
nx, nu, N, legs = 36, 12, 256, 8
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=TRACE_LIB)
s.upload([prob]); s.backward(1e-12)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-12)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))
seq = [(0, "start"), (1, "vplus"), (2, "qhat"), (3, "P,H,Ghat_u"), (4, "export+Bop"), (5, "factor"), (11, "G read"),
       (12, "subst x2"), (6, "write+sync"), (7, "Kb,Kth,yff,vx,vt"),
       (8, "Aff,Yth,Vxt,Vtt,load_a"), (9, "Vxx"), (10, "store")]
prev = t[0]
print(s.kernel_name, "stage total", t[10] - t[0])
for k, nm in seq[1:]:
    print(f"  {nm:18s} {t[k]-prev:7d}"); prev = t[k]

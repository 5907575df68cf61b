"""Phase-by-phase cycle breakdown of one parameterised stage of the wave-leg backward kernel."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, legs = 36, 12, 256, 8
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs)
s.upload([prob]); s.backward(1e-12)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-12)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))
seq = [(0, "start"), (1, "vplus"), (2, "qhat"), (3, "P,H,Ghat_u"), (4, "export+Bop"), (5, "factor"), (11, "G read"),
       (12, "subst x2"), (6, "write+sync"), (7, "Kb,Kth,yff,vx,vt"), (16, "Aff+Yth"), (17, "Vxt"), (18, "Vtt"),
       (8, "load_a"), (9, "Vxx"), (10, "store")]
prev = t[0]
print(s.kernel_name, "stage total", t[10] - t[0])
for k, nm in seq[1:]:
    print(f"  {nm:18s} {t[k]-prev:7d}"); prev = t[k]

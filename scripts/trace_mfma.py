"""Phase-by-phase cycle breakdown of one stage of the specialised backward kernel."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=B)
synth_device.fill_problems(s, seed=1, mode="W")
s.backward(1e-14)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-14)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out)).reshape(4, 16)
names = ["start", "S1|vp", "S2|h", "export|pf", "barA", "barB|factor", "Aff|solve", "Vxx|barB", "barC|tail", "store|barC", "-"]
base = t[:, 0].min()
print(f"batch {B}: cycle stamps relative to stage start (s_memtime ticks)")
for wv in range(4):
    print("wave", wv, " ".join(f"{names[i]}={t[wv, i] - base}" for i in range(11) if t[wv, i]))

"""Phase-by-phase cycle breakdown of one stage of the one-wave-per-problem backward kernel."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")  # make -C aligator_amd/csrc trace
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=B, lib_path=TRACE_LIB)
synth_device.fill_problems(s, seed=1, mode="W")
s.backward(1e-14)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-14)
s._L.gar_hip_debug_trace(s.handle, 0, out)
tt = np.array(list(out)); t = tt[:11]; print("init: factor", tt[14]-tt[13], "solve", tt[15]-tt[14]); print("solve sub-marks: G-read", tt[11]-tt[5], "subst", tt[12]-tt[11], "write+sync", tt[6]-tt[12])
names = ["start", "vplus", "qhat", "S1S2", "export", "factor", "solve", "Kb+vec", "Aff", "Vxx", "store"]
print(f"{s.kernel_name} batch {B}: cycles per phase (s_memtime ticks), total {t[10]-t[0]}")
print(" ".join(f"{names[i]}={t[i]-t[i-1]}" for i in range(1, 11)))

"""Per-phase cycle stamps of both waves of the two-waves-per-problem stage (gar_wave_pair.hpp); needs the
tracing build (make -C aligator_amd/csrc trace)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nx, nu, N = 56, 24, 64
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=B, lib_path=LIB)
pk = s.pack(synth.generate_lq_problem(3, np.ones(nx), N, nx, nu, mode="W"))
for b in range(B):
    s.upload_packed(pk, b, 1)
s.backward(1e-12)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-12)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out)).reshape(4, 16)
names = ["P,H own cols", "hq,export", "wait(1)", "factor (w1)", "wait(2)", "solve", "wait(3)", "K st,Bop,kff,yff,vx", "Aff", "Vxx,V->LDS",
         "next-knot loads", "wait(4)", "flush"]
print(f"{s.kernel_name} batch {B}: stage total {t[0][13]-t[0][0]} cycles")
for w in (0, 1):
    print(f" wave {w}: " + " | ".join(f"{names[i]}={t[w][i+1]-t[w][i]}" for i in range(13)))

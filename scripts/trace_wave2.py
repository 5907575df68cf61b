"""Phase-by-phase cycle breakdown of one stage of the second-generation plain stage (gar_wave2.hpp)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")  # make -C aligator_amd/csrc trace
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nx, nu = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (36, 12)
nc = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # nc > 0: the decoupled constrained stage (generator: D = 0)
N = 256 if nx <= 36 and nc == 0 else 64
dims = [(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=B, lib_path=TRACE_LIB)
if nx <= 36 and nc == 0:
    synth_device.fill_problems(s, seed=1, mode="W")
else:
    from aligator_amd import synth
    import numpy as np
    pk = s.pack(synth.generate_lq_problem(3, np.ones(nx), N, nx, nu, nc=nc, mode="W"))
    for b in range(B):
        s.upload_packed(pk, b, 1)
s.backward(1e-11 if nc else 1e-14)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-11 if nc else 1e-14)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))
marks = [(0, "start"), (1, "P,H col 2"), (2, "P,H cols 1,0 + slots A"), (3, "factor"), (4, "hq"),
         (5, "solve-operands"), (6, "solve"), (7, "kff,yff,vx"), (11, "Aff col 0 (+K stores)"), (12, "Aff col 1 (+col 0 st/ld)"), (8, "Aff col 2 (+col 1 st/ld)"), (13, "Vxx rem4 tiles"), (15, "Vxx 16x16 tiles + slots"), (14, "drain + sync"), (9, "load_b"), (10, "end")]
print(f"{s.kernel_name} batch {B}: cycles per phase (s_memtime ticks), total {t[10]-t[0]}")
print(" | ".join(f"{marks[i][1]}={t[marks[i][0]]-t[marks[i-1][0]]}" for i in range(1, len(marks))))

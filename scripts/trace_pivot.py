"""What a stage costs when Bunch-Kaufman really pivots on Rhat (the plain stage's out-of-line path): every stage of
these problems interchanges at column 0 (first control without dynamics, tiny R(0,0) against R(1,0) = 1).  Needs the
tracing build; GAR_LIB selects a library build (timing experiments)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("GAR_LIB") or os.path.join(ROOT, "aligator_amd", "libgar_hip_trace.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nx, nu, N = 36, 12, 64
os.environ["GAR_HIP_BACKWARD"] = "wave"
prob = synth.generate_lq_problem(3, np.ones(nx), N, nx, nu, mode="W")
for k in prob.stages[:-1]:
    k.B[:, 0] = 0.0
    k.S[:, 0] = 0.0
    k.R[0, :] = 0.0; k.R[:, 0] = 0.0
    k.R[0, 0] = 1e-6; k.R[1, 0] = k.R[0, 1] = 1.0
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=B, lib_path=LIB)
pk = s.pack(prob)
for b in range(B):
    s.upload_packed(pk, b, 1)
s.backward(1e-12)
print("pivoted stage fraction:", s.slow_path_stages()[1] / (N * B), "failed:", s.num_failed())
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-12)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))
print(f"{s.kernel_name} batch {B}: stage {t[10]-t[0]} cycles; factor (register LDL^T attempt) {t[3]-t[2]}; "
      f"hq {t[4]-t[3]}; export + out-of-line Bunch-Kaufman + solve {t[6]-t[4]}")

"""Phase cycle stamps of one constrained stage (nx=36, nu=12, nc=32) of the wave kernel (debug build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
TRACE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligator_amd", "libgar_hip_trace.so")
nx, nu, nc, N = 36, 12, 32, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
if os.environ.get("DNONZERO"):   # the reference's generator leaves D = 0 (test_util.cpp:42-43)
    rng = np.random.default_rng(9)
    for k in prob.stages[:-1]:
        k.D[...] = rng.uniform(-1, 1, k.D.shape)
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=B, lib_path=TRACE_LIB)
pk = s.pack(prob)
for b in range(B):
    s.upload_packed(pk, b, 1)
s.backward(1e-8)
out = (C.c_longlong * 64)()
s._L.gar_hip_debug_trace(s.handle, 1, None)
s.backward(1e-8)
s._L.gar_hip_debug_trace(s.handle, 0, out)
t = np.array(list(out))[:11]
names = ["start", "vplus", "qhat", "S1S2", "export", "KKT assemble+factor", "KKT solve (37 rhs)", "Kb+vec+Z out", "Aff", "Vxx+C^TZ", "store"]
print(s.kernel_name, "batch", B, "D random" if os.environ.get("DNONZERO") else "D = 0", "stage total", t[10] - t[0])
prev = t[0]
for i in range(1, 11):
    if t[i]:
        print(f"  {names[i]:18s} {t[i]-prev:8d}"); prev = t[i]

tt = np.array(list(out))
if tt[20]:
    print(f"  coupled stage: X init {tt[20]-tt[5]}, packed MFMA solve {tt[21]-tt[20]}, C reload issue {tt[22]-tt[21]}, to mark 6 {tt[6]-tt[22]}")

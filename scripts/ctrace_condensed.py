"""Cycles per phase of gar_condensed_generic (workgroup 0) on the Talos-walk LQ shape in leg mode.
Needs the debug build:  make -C aligator_amd/csrc ctrace   (libgar_hip_ctrace.so, -DGAR_CTRACE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
LIB = os.path.join(ROOT, "aligator_amd", "libgar_hip_ctrace.so")
nx, nu, N = 56, 22, 275
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
names = ["prologue", "assemble", "store blk", "factor", "build ublk", "solve", "store U", "stage next", "gemm", "-",
         "back-subst", "refine"]
for legs in (8, 16):
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=LIB)
    s.upload([prob])
    for _ in range(2):
        s.backward_async(1e-10); s.forward_async()
    s.sync()
    lib = C.CDLL(LIB)
    out = (C.c_longlong * 16)()
    lib.gar_hip_debug_ctrace(out)
    s.backward_async(1e-10); s.forward_async(); s.sync()
    lib.gar_hip_debug_ctrace(out)
    print(f"legs={legs}: total {sum(out)} cycles;", {n: int(out[i]) for i, n in enumerate(names)}, flush=True)
    s.close()

"""Cycles per phase of the cyclic-reduction kernels of the reduced condensed system (gar_condensed_cr.hpp), workgroup
(0, 0) of each launch, Talos-walk LQ shape in leg mode.  Needs  make -C aligator_amd/csrc ctrace  (-DGAR_CTRACE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
LIB = os.path.join(ROOT, "aligator_amd", "libgar_hip_ctrace.so")
nx, nu, N = 56, 22, 275
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
names = ["elim: load", "elim: factor", "elim: solve", "elim: store", "update: right", "update: left", "-",
         "back: load", "back: factor", "back: solve", "back: levels", "back: store", "assemble: block 0"]
for legs in (int(a) for a in (sys.argv[1:] or ["16"])):
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=LIB)
    s.upload([prob])
    for _ in range(2):
        s.backward_async(1e-10); s.forward_async()
    s.sync()
    lib = C.CDLL(LIB)
    out = (C.c_longlong * 16)()
    lib.gar_hip_debug_crtrace(out)
    s.backward_async(1e-10); s.forward_async(); s.sync()
    lib.gar_hip_debug_crtrace(out)
    print(f"legs={legs} ({s.condensed_solver_name}; the eliminate / update figures are sums over the levels):",
          {n: int(out[i]) for i, n in enumerate(names) if n != "-"}, flush=True)
    s.close()

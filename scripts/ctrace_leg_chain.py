"""Cycles per phase of gar_leg_param_chain (leg 1, thread 0) on the Talos-walk LQ shape, N = 256, 32 legs.
Needs the debug build:  make -C aligator_amd/csrc ctrace   (libgar_hip_ctrace.so, -DGAR_CTRACE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
LIB = os.path.join(ROOT, "aligator_amd", "libgar_hip_ctrace.so")
nx, nu, N, legs = 56, 22, 256, 32
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=LIB)
s.upload([prob])
lib = C.CDLL(LIB)
out = (C.c_longlong * 16)()
for _ in range(3):
    s.backward_async(1e-10); s.sync()
lib.gar_hip_debug_ptrace(out)
reps = 10
for _ in range(reps):
    s.backward_async(1e-10); s.sync()
lib.gar_hip_debug_ptrace(out)
names = ["loop tail -> barrier", "top barrier", "descriptor + prefetch issue", "product", "mid barrier", "operand commit (waits for the prefetch)",
         "descriptor to scalars", "record store issue"]
stages = N // legs
print(f"gar_leg_param_chain, {stages} stages per leg, cycles per STAGE:", {n: int(out[i]) // (reps * stages) for i, n in enumerate(names)},
      "sum", sum(out[:8]) // (reps * stages), flush=True)

"""One (36, 12, nc = 32) problem with a random D on every knot, N = 256, in leg mode (LEGS, default 32): 20 sweeps -- the
process `rocprofv3 --kernel-trace --stats` is pointed at for the kernel split of the constrained segment legs
(csrc/gar_cstr_seg.hpp; GAR_HIP_CSTR_SEG_LEGS=0: the any-dimension leg kernels)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, nc, N, mu = 36, 12, 32, 256, 1e-8
legs = int(os.environ.get("LEGS", "32"))
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
rng = np.random.default_rng(9)
for k in prob.stages[:-1]:
    k.D[...] = rng.uniform(-1, 1, k.D.shape)
LIB = os.environ.get("LIB")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=(os.path.join(ROOT, "aligator_amd", LIB) if LIB else None))
s.upload([prob])
import time
for _ in range(5):
    s.backward_async(mu); s.forward_async()
s.sync()
t0 = time.perf_counter()
for _ in range(20):
    s.backward_async(mu); s.forward_async()
s.sync()
print(s.kernel_name, LIB or "libgar_hip.so", f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per sweep, {legs} legs", "done")

"""One constrained problem (36, 12, 32), N = 256, `legs` legs, swept a few times: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, nc, N = 36, 12, 32, int(os.environ.get("N", "256"))
legs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs)
s.upload([prob])
for _ in range(3):
    s.backward_async(1e-11); s.forward_async()
s.sync()
t0 = time.perf_counter()
for _ in range(10):
    s.backward_async(1e-11); s.forward_async()
s.sync()
print(s.kernel_name, "legs", legs, "ms per sweep", (time.perf_counter() - t0) / 10 * 1e3)

"""One parallel-in-time sweep configuration for rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu = 36, 12
N = int(os.environ.get("HORIZON", "2048")); legs = int(os.environ.get("LEGS", "64")); batch = int(os.environ.get("BATCH", "1"))
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=batch, num_legs=legs)
s.upload([prob] * batch)
for _ in range(20):
    s.backward_async(1e-12); s.forward_async()
s.sync()
print("done", s.kernel_name)

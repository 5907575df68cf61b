import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, mueq = 36, 12, 256, 1e-12
legs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = synth.generate_lq_problem(5, np.zeros(nx), N, nx, nu, mode="W")
dims = [(a, b, c, d, 0) for (a, b, c, d, _) in (k.dims for k in prob.stages)]
s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs)
s.upload([prob])
for _ in range(5):
    s.backward_async(mueq); s.forward_async(); s.sync()

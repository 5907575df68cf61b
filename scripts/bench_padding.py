"""What padding onto a specialised shape buys (aligator_amd/gar.py::_padded_dims): sweeps/s of shapes
without a kernel of their own, padded (GAR_HIP_PAD=1, default) against the generic kernels
(GAR_HIP_PAD=0), N=256, batch 1024, data resident."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver

N, batch, mu = 256, 1024, 1e-12
for nx, nu in ((30, 10), (13, 5), (10, 3), (4, 2)):
    probs = [synth.generate_lq_problem(7 + i, np.zeros(nx), N, nx, nu, mode="W") for i in range(2)]
    for pad in ("1", "0"):
        os.environ["GAR_HIP_PAD"] = pad
        s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], probs[0].nc0, batch=batch)
        s.upload([probs[b % 2] for b in range(batch)])
        s.backward(mu); s.forward(); s.sync()
        reps = 5 if pad == "1" else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            s.backward_async(mu)
            s.forward_async()
        s.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"nx={nx:2d} nu={nu:2d} pad={pad} {s.kernel_name:14s} {dt * 1e3:9.2f} ms  {batch / dt:10.0f} sweeps/s", flush=True)
        s.close()

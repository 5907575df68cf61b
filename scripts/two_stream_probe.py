"""Does the forward sweep (memory-bound) hide behind the backward sweep (issue-bound) of ANOTHER half of the batch?
One solver of 4096 problems on one stream against two solvers of 2048 on two streams, same total work per step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
def make(batch):
    s = BatchedRiccatiSolver(dims, nx, batch=batch)
    st = torch.cuda.Stream()
    s.set_stream(st.cuda_stream)
    synth_device.fill_problems(s, seed=1, mode="W", keep=())
    return s, st
def run(solvers, steps=10):
    for s, _ in solvers:
        for _ in range(2):
            s.backward_async(1e-14); s.forward_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for s, _ in solvers:
            s.backward_async(1e-14)
        for s, _ in solvers:
            s.forward_async()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
def run_staggered(solvers, steps=10):
    a, b = solvers[0][0], solvers[1][0]
    for s in (a, b):
        for _ in range(2):
            s.backward_async(1e-14); s.forward_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.backward_async(1e-14)
    for _ in range(steps):
        b.backward_async(1e-14); a.forward_async()
        a.backward_async(1e-14); b.forward_async()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (steps + 0.5)
one = [make(4096)]
t1 = run(one)
print(f"one solver of 4096, one stream:      {t1*1e3:7.3f} ms/step  {4096/t1:9.0f} sweeps/s")
del one; torch.cuda.empty_cache()
two = [make(2048), make(2048)]
t2 = run(two)
print(f"two solvers of 2048, two streams:    {t2*1e3:7.3f} ms/step  {4096/t2:9.0f} sweeps/s")
t3 = run_staggered(two)
print(f"  ... staggered (fwd(A) with bwd(B)): {t3*1e3:7.3f} ms/step  {4096/t3:9.0f} sweeps/s")

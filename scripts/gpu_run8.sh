#!/bin/bash
set -u
export TMPDIR=/tmp
for B in 16 64 256 1024; do
python - <<PY
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, B = 36, 12, 256, $B
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
for legs in (1, 2, 4, 8, 16):
    if B * legs > 8192: continue
    s = BatchedRiccatiSolver(dims, nx, batch=B, num_legs=legs)
    synth_device.fill_problems(s, seed=7, mode="W", keep=())
    for _ in range(2):
        s.backward_async(1e-14); s.forward_async()
    s.sync()
    t0 = time.perf_counter(); R = 5
    for _ in range(R):
        s.backward_async(1e-14); s.forward_async()
    s.sync()
    dt = (time.perf_counter() - t0) / R
    print(f"N={N} batch={B:5d} legs={legs:3d} {s.kernel_name:16s} {dt*1e3:8.3f} ms/step  {B/dt:10.0f} sweeps/s", flush=True)
PY
done

"""A/B on ONE box, alternating launches: the serial (36, 12) sweep with the packed lower triangle of Vxx in its
factor records (libgar_hip.so) against the full block (libgar_hip_vxxfull.so: make -C aligator_amd/csrc vxxfull).
Backward and forward kernel times from the library's HIP events, batch 4 096, N = 256."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N, batch = 36, 12, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
libs = {"packed": os.path.join(ROOT, "aligator_amd", "libgar_hip.so"),
        "full": os.path.join(ROOT, "aligator_amd", "libgar_hip_vxxfull.so")}
if len(sys.argv) > 2:   # any two builds: name=path name=path
    libs = {a.split("=")[0]: os.path.join(ROOT, "aligator_amd", a.split("=")[1]) for a in sys.argv[2:]}
solvers = {}
for name, path in libs.items():
    s = BatchedRiccatiSolver(dims, nx, batch=batch, num_legs=1, device=0, lib_path=path)
    synth_device.fill_problems(s, seed=1234, mode="W", keep=())
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    for _ in range(2):
        s.backward_async(1e-14); s.forward_async()
    s.sync()
    solvers[name] = s
times = {k: [] for k in solvers}
for rep in range(8):
    for name, s in solvers.items():
        s.backward_async(1e-14); s.forward_async(); s.sync()
        o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
        times[name].append((o[0], o[2]))
for name, t in times.items():
    a = np.array(t)
    print(f"{name:7s} backward median {np.median(a[:, 0]):.3f} (min {a[:, 0].min():.3f}, max {a[:, 0].max():.3f})  "
          f"forward median {np.median(a[:, 1]):.3f} (min {a[:, 1].min():.3f})  step {np.median(a.sum(1)):.3f} ms "
          f"=> {batch / np.median(a.sum(1)) * 1e3:.0f} sweeps/s")
# same answers
x = [solvers[k].solution(0) for k in solvers]
print("max difference between the builds over the solution of problem 0:",
      max(float(np.abs(a - b).max()) for y in x[1:] for A, B in zip(x[0], y) for a, b in zip(A, B) if a.size))

"""A/B of builds of the library on ONE box, alternating launches, on ONE secondary shape with every problem of the batch
its own device-generated draw (aligator_amd/synth_device.py): SHAPE=talos (56, 22, N=275; default), nc32, nc32c (coupled
D != 0), north (36, 12, N=256).  Backward / forward kernel times from the library's HIP events, roofline fraction of the
backward sweep on the shape's algorithmic bytes, and the solutions of the builds against each other.
usage: SHAPE=talos BATCH=1024 python scripts/ab_shape.py name=libgar_hip.so name=libgar_hip_pairbase.so ..."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
libs = {a.split("=")[0]: os.path.join(ROOT, "aligator_amd", a.split("=")[1]) for a in sys.argv[1:]}
SH = {"talos": (56, 22, 0, 275, 1e-10, False), "nc32": (36, 12, 32, 256, 1e-11, False), "nc32c": (36, 12, 32, 256, 1e-11, True),
      "north": (36, 12, 0, 256, 1e-14, False)}
shape = os.environ.get("SHAPE", "talos")
nx, nu, nc, N, mu, coupled = SH[shape]
batch = int(os.environ.get("BATCH", "1024"))
dims = [(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)]
solvers = {}
for name, path in libs.items():
    s = BatchedRiccatiSolver(dims, nx, batch=batch, device=0, lib_path=path)
    synth_device.fill_problems(s, seed=11, mode=os.environ.get("GEN", "W"), coupled=coupled)
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    for _ in range(2):
        s.backward_async(mu); s.forward_async()
    s.sync()
    solvers[name] = s
times = {k: [] for k in solvers}
for rep in range(int(os.environ.get("REPS", "6"))):
    for name, s in solvers.items():
        s.backward_async(mu); s.forward_async(); s.sync()
        o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
        times[name].append((o[0], o[2]))
knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu + nc * (nx + nu + 1)
fac = (nu + nc + nx) * (nx + 1) + nx * nx + nx
for name, t in times.items():
    a = np.array(t)
    bw = np.median(a[:, 0])
    print(f"{shape} ({nx},{nu},nc={nc}) batch {batch} {solvers[name].kernel_name:16s} {name:10s} backward median {bw:.3f} (min {a[:, 0].min():.3f}) "
          f"= {8 * (knot + fac) * N * batch / (bw * 1e-3) / 8e12:.3f} of the HBM roofline; forward {np.median(a[:, 1]):.3f}; step {np.median(a.sum(1)):.3f} ms "
          f"=> {batch / np.median(a.sum(1)) * 1e3:.0f} sweeps/s  failed {solvers[name].num_failed()}", flush=True)
names = list(solvers)
x = [[solvers[k].solution(b) for b in (0, batch - 1)] for k in names]
for k, xi in zip(names[1:], x[1:]):
    d = 0.0
    for pa, pb in zip(x[0], xi):
        sc = max(1.0, max(float(np.abs(v).max()) for part in pa for v in part if v.size))
        d = max(d, max(float(np.abs(a - b).max()) for A, B in zip(pa, pb) for a, b in zip(A, B) if a.size) / sc)
    print(f"   max relative difference {names[0]} vs {k} (problems 0 and {batch - 1}): {d:.2e}", flush=True)
for s in solvers.values():
    s.close()

"""A/B of two builds of the library on ONE box, alternating launches, three shapes of the serial hot path: the north
star (36, 12, N=256, batch 4 096), the Talos-walk LQ shape (56, 22, N=275, batch 1 024) and the reference's own
benchmark shape (36, 12, nc=32, N=256, batch 1 024).  Backward / forward kernel times from the library's HIP events.
usage: ab_libs.py name=libA.so name=libB.so   (paths relative to aligator_amd/)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
libs = {a.split("=")[0]: os.path.join(ROOT, "aligator_amd", a.split("=")[1]) for a in sys.argv[1:]}
SHAPES = ((36, 12, 0, 256, 4096, 1e-14), (56, 22, 0, 275, 1024, 1e-10), (36, 12, 32, 256, 1024, 1e-11))
if os.environ.get("AB_ONLY"):   # e.g. AB_ONLY=0: the north star alone
    SHAPES = tuple(SHAPES[int(i)] for i in os.environ["AB_ONLY"].split(","))
for nx, nu, nc, N, batch, mu in SHAPES:
    probs = [synth.generate_lq_problem(100 + i, np.zeros(nx), N, nx, nu, nc=nc, mode="W") for i in range(2)]
    solvers = {}
    for name, path in libs.items():
        s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch, device=0, lib_path=path)
        packed = np.concatenate([s.pack(p) for p in probs])
        for b0 in range(0, batch, 2):
            s.upload_packed(packed, b0, 2)
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        for _ in range(2):
            s.backward_async(mu); s.forward_async()
        s.sync()
        solvers[name] = s
    times = {k: [] for k in solvers}
    for rep in range(6):
        for name, s in solvers.items():
            s.backward_async(mu); s.forward_async(); s.sync()
            o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
            times[name].append((o[0], o[2]))
    for name, t in times.items():
        a = np.array(t)
        print(f"({nx},{nu},nc={nc}) batch {batch} {solvers[name].kernel_name:16s} {name:9s} backward median {np.median(a[:, 0]):.3f} "
              f"(min {a[:, 0].min():.3f})  forward median {np.median(a[:, 1]):.3f}  step {np.median(a.sum(1)):.3f} ms "
              f"=> {batch / np.median(a.sum(1)) * 1e3:.0f} sweeps/s  failed {solvers[name].num_failed()}", flush=True)
    x = [solvers[k].solution(0) for k in solvers]
    sc = max(1.0, max(float(np.abs(v).max()) for part in x[0] for v in part if v.size))
    for name, xi in zip(list(solvers)[1:], x[1:]):
        print(f"   max relative difference {list(solvers)[0]} vs {name}, problem 0:",
              max(float(np.abs(a - b).max()) for A, B in zip(x[0], xi) for a, b in zip(A, B) if a.size) / sc, flush=True)
    for s in solvers.values():
        s.close()

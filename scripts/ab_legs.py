"""A/B of builds of the library on ONE box in LEG mode, one problem (the seam's configuration): (36, 12), N = 256, 32 legs
(and SHAPE=talos: (56, 22)); per-phase device times from the library's own HIP events (leg sweep | condensed solve |
roll-out), alternating between the builds; the solutions against each other and the condensed solver's own verdict.
usage: [SHAPE=north|talos] [LEGS=32] python scripts/ab_legs.py name=libgar_hip.so name=libgar_hip_noblk.so"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
libs = {a.split("=")[0]: os.path.join(ROOT, "aligator_amd", a.split("=")[1]) for a in sys.argv[1:]}
nx, nu = (56, 22) if os.environ.get("SHAPE") == "talos" else (36, 12)
if os.environ.get("NX"):
    nx, nu = int(os.environ["NX"]), int(os.environ.get("NU", "12"))
N, legs, mu = 256, int(os.environ.get("LEGS", "32")), 1e-10
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
solvers = {}
for name, path in libs.items():
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=path)
    s.upload([prob])
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    for _ in range(3):
        s.backward_async(mu); s.forward_async()
    s.sync()
    solvers[name] = s
acc = {k: [] for k in solvers}
for rep in range(20):
    for name, s in solvers.items():
        s.backward_async(mu); s.forward_async(); s.sync()
        o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
        acc[name].append(list(o))
for name, s in solvers.items():
    a = np.array(acc[name]) * 1e3
    res, steps = s.condensed_info(0)
    print(f"({nx},{nu}) N={N} legs={legs} {s.kernel_name:18s} {name:8s} leg sweep {np.median(a[:, 0]):7.1f}  condensed {np.median(a[:, 1]):7.1f}  roll-out {np.median(a[:, 2]):6.1f} us"
          f"  (min {a[:, 0].min():.1f} / {a[:, 1].min():.1f} / {a[:, 2].min():.1f})  condensed solver {s.condensed_solver_name}, redone by the chain: {bool(s.condensed_resolved(0))}, "
          f"residual {res:.1e}, omega {s.condensed_backward_error(0):.1e}", flush=True)
names = list(solvers)
x0 = solvers[names[0]].solution(0)
sc = max(1.0, max(float(np.abs(v).max()) for part in x0 for v in part if v.size))
for k in names[1:]:
    xi = solvers[k].solution(0)
    print(f"   max relative difference {names[0]} vs {k}: {max(float(np.abs(a - b).max()) for A, B in zip(x0, xi) for a, b in zip(A, B) if a.size) / sc:.2e}")

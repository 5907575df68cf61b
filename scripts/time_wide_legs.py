"""The Talos-walk LQ shape (56, 22), N = 275, ONE problem: serial (pair<56,24>) and PARALLEL with 2 ... 34 legs
(pair_leg<56,24>, gar_leg_seg.hpp) -- per-phase times from the library's HIP events and the wall time per sweep."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 56, 22, 275
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
ref = None
# (legs, GAR_HIP_CONDENSED_CR): from 4 legs on the reduced condensed system goes through block cyclic reduction
# (gar_condensed_cr.hpp); "0" = the one-workgroup chain on the same reduced system, for the A/B on one box
runs = [(1, None), (2, None)] + [(J, cr) for J in (4, 5, 8, 16, 34, 68) for cr in ("0", None)]
if os.environ.get("LEGS"):   # one configuration (for rocprofv3 --kernel-trace --stats: scripts/prof_wide_legs.sh)
    runs = [(int(os.environ["LEGS"]), None)]
for legs, cr in runs:
    if cr is None:
        os.environ.pop("GAR_HIP_CONDENSED_CR", None)
    else:
        os.environ["GAR_HIP_CONDENSED_CR"] = cr
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs)
    s.upload([prob])
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    for _ in range(2):
        s.backward_async(1e-10); s.forward_async()
    s.sync()
    acc = np.zeros(3); t0 = time.perf_counter()
    for _ in range(5):
        s.backward_async(1e-10); s.forward_async()
        o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o)); acc += np.array(list(o))
    wall = (time.perf_counter() - t0) / 5 * 1e3
    sol = s.solution(0)
    if ref is None:
        ref = sol
    err = max(float(np.abs(a - b).max()) for A, B in zip(sol, ref) for a, b in zip(A, B) if a.size) / max(1.0, max(float(np.abs(v).max()) for v in ref[3]))
    cond = f"{s.condensed_solver_name:14s} resolved={int(s.condensed_resolved(0))}" if legs > 1 else " " * 25
    print(f"legs={legs:3d} {s.kernel_name:16s} {cond} backward {acc[0]/5:.3f} condensed/initial {acc[1]/5:.3f} forward {acc[2]/5:.3f} ms  wall {wall:.3f} ms  |sol - serial| {err:.1e}", flush=True)
    s.close()

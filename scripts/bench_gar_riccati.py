"""The reference's own gar benchmark (bench/gar-riccati.cpp:19-90) on the MI355X backend: nx = 36, nu = 12,
nc = 32 constraints on every knot (generator: C = [I 0], D = 0), mu = 1e-11, N = 2^4 .. 2^10, ONE problem
per call as the reference times it, in milliseconds per backward + forward:
  BM_serial      ProximalRiccatiSolver      -> constrained wave kernels (wave<36,12,32>)
  BM_stagedense  RiccatiSolverDense         -> gar_dense.hpp
  BM_parallel<J> ParallelRiccatiSolver, J in {2, 3, 4, 6} legs (and N/8 legs: what a GPU wants)
beside the restated reference (the oracle, one thread = BM_serial's role) on the GPU box's host.
A second table: the same sweeps at batch 1024 (sweeps/s), which is what the hardware is for."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
from oracle import oracle as ora

nx, nu, nc, mueq = 36, 12, 32, 1e-11
ora.lib(native=True)


def gpu_ms(dims, prob, batch, reps, **kw):
    try:
        s = BatchedRiccatiSolver(dims, nx, batch=batch, **kw)
    except RuntimeError as e:
        return None, "refused: " + str(e)[:60]
    pk = s.pack(prob)
    for b in range(batch):
        s.upload_packed(pk, b, 1)
    s.backward(mueq); s.forward(); s.sync()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        s.backward_async(mueq); s.forward_async(); s.sync()
        best = min(best, time.perf_counter() - t0)
    name = s.kernel_name
    s.close()
    return best * 1e3, name


print(f"{'N':>5s} | {'oracle 1 thr':>12s} | {'serial':>9s} | {'stagedense':>10s} | " + " | ".join(f"{'par J=' + str(j):>9s}" for j in (2, 3, 4, 6)) + f" | {'par N/8':>9s}   (ms per backward+forward, one problem)")
names = {}
for e in range(4, 11):
    N = 1 << e
    prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
    dims = [k.dims for k in prob.stages]
    op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0, native=True)
    so = ora.ProximalRiccatiSolver(op)
    sol = op.initialize_solution()
    t_or = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        so.backward(mueq); so.forward(*sol)
        t_or = min(t_or, time.perf_counter() - t0)
    row = [f"{N:5d}", f"{t_or * 1e3:12.3f}"]
    for key, kw in (("serial", {}), ("stagedense", {"dense": True})):
        ms, nm = gpu_ms(dims, prob, 1, 5, **kw)
        names[key] = nm
        row.append(f"{ms:{9 if key == 'serial' else 10}.3f}" if ms is not None else nm)
    for j in (2, 3, 4, 6, max(2, N // 8)):
        ms, nm = gpu_ms(dims, prob, 1, 5, num_legs=j)
        names["parallel"] = nm
        row.append(f"{ms:9.3f}" if ms is not None else f"{'-':>9s}")
    print(" | ".join(row), flush=True)
print("kernels:", names)

print("\nbatch 1024, N = 256 (sweeps/s):")
N = 256
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
dims = [k.dims for k in prob.stages]
for key, kw in (("serial", {}), ("stagedense", {"dense": True}), ("parallel J=4", {"num_legs": 4}), ("parallel J=32", {"num_legs": 32})):
    ms, nm = gpu_ms(dims, prob, 1024, 3, **kw)
    print(f"  {key:14s} {nm:18s} " + (f"{1024 / (ms * 1e-3):10.0f} sweeps/s" if ms is not None else ""), flush=True)

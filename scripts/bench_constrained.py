"""The reference's own benchmark shape (bench/gar-riccati.cpp: nx=36, nu=12, nc=32 on every knot,
N = 2^e) on the generic kernels: batched sweeps/s (secondary figure; BASELINE.json's metric is the
unconstrained north star)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver, lqrComputeKktError

nx, nu, nc, N, mueq = 36, 12, 32, int(os.environ.get("HORIZON", "256")), 1e-11
probs = [synth.generate_lq_problem(100 + i, np.zeros(nx), N, nx, nu, nc=nc, mode="W") for i in range(2)]
if os.environ.get("DNONZERO"):   # the reference's generator leaves D = 0 (test_util.cpp:42-43)
    for p in probs:
        rng = np.random.default_rng(9)
        for k in p.stages[:-1]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
dims = [k.dims for k in probs[0].stages]
for B in (64, 256, 1024, 4096):
    s = BatchedRiccatiSolver(dims, nx, batch=B)
    packed = np.concatenate([s.pack(p) for p in probs])
    for b0 in range(0, B, 2):
        s.upload_packed(packed, b0, 2)
    s.backward(mueq); s.forward(); s.sync()
    kkt = max(lqrComputeKktError(probs[1], *s.solution(B - 1), mueq=mueq))
    t0 = time.perf_counter(); R = 3
    for _ in range(R):
        s.backward_async(mueq); s.forward_async()
    s.sync()
    dt = (time.perf_counter() - t0) / R
    knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu + nc * (nx + nu + 1)      # Q S R q r A B f + C D d
    fac = (nu + nc + nx) * (nx + 1) + nx * nx + nx                                   # [K; Z; Aff | kff; zff; yff], Vxx, vx
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    s.backward_async(mueq); s.forward_async(); s.sync()
    import ctypes as C
    o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
    frac = 8 * (knot + fac) * N * B / (o[0] * 1e-3) / 8e12
    first, piv = s.slow_path_stages()
    print(("D random " if os.environ.get("DNONZERO") else "D = 0    ") + f"nc={nc} N={N} batch={B:5d} {s.kernel_name:10s} {dt*1e3:9.2f} ms/step {B/dt:9.0f} sweeps/s  kkt {kkt:.1e}"
          f"  backward kernels {o[0]:.2f} ms = {frac:.3f} of 8 TB/s; stages: {s.constrained_bk_stages()[0] / (N * B):.3f} coupled (register LDL^T of the 44 x 44), {s.constrained_bk_stages()[1] / (N * B):.3f} on the LDS Bunch-Kaufman,"
          f" {first / (N * B):.3f} needed the second Bunch-Kaufman test", flush=True)

# the CPU oracle (restated reference, oracle/gar_oracle.c -O3 -march=native, OpenMP over problems) on
# this box's host cores, same shape and data
from oracle import oracle as ora
ora.lib(native=True)
cores = os.cpu_count() or 1
many = [ora.Problem.from_knots(p.stages, p.G0, p.g0, native=True) for p in (probs * cores)[:2 * cores]]
bs = ora.BatchSweep(many)
threads = bs.max_threads()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import cpu_quota_cores   # (the pool's boxes: 256 logical CPUs, a 16-CPU cgroup quota)
if cpu_quota_cores() is not None:
    threads = max(1, min(threads, cpu_quota_cores()))
bs.sweep(mueq, threads)
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 8.0:
    assert bs.sweep(mueq, threads) == 0
    reps += 1
dt = time.perf_counter() - t0
print(f"CPU oracle nc={nc} N={N}: {len(many) * reps / dt:8.0f} sweeps/s on {threads} threads", flush=True)


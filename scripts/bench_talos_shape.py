"""BASELINE.json configs[4]'s LQ sub-problem shape (bench/talos-walk.cpp:20-28: nx = 56, nu = 22, N = 275;
bench/lqr.cpp:25-57's blocks) on the one-wave-per-problem backward kernel wave<56,24> (controls padded), generic
forward: sweeps/s over the batch and the backward kernel's fraction of the HBM roofline, against the
ALGORITHMIC bytes of the caller's (56, 22) shape."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 56, 22, 275
knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu
fac = (nu + nx) * nx + (nu + nx) + nx * nx + nx
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
probs = [synth.generate_lq_problem(900 + i, np.ones(nx), N, nx, nu, mode="W") for i in range(4)]
for batch in [int(a) for a in sys.argv[1:]] or [256, 768, 1536]:
    for force, wide in (("0", "pair"), ("0", "single"), ("1", "pair")):
        if force == "1" and batch > 256:
            continue
        if os.environ.get("GAR_LIB") and (force, wide) != ("0", "pair"):
            continue  # A/B runs of kernel variants: the pair kernel only
        os.environ["GAR_HIP_FORCE_GENERIC"] = force
        os.environ["GAR_HIP_WIDE"] = wide
        os.environ["GAR_HIP_PAD"] = "1" if force == "0" else "0"
        s = BatchedRiccatiSolver(dims, nx, batch=batch, lib_path=os.environ.get("GAR_LIB") or None)
        packed = np.concatenate([s.pack(p) for p in probs])
        for b0 in range(0, batch, 4):
            s.upload_packed(packed[: min(4, batch - b0) * s.problem_doubles], b0, min(4, batch - b0))
        s.backward(1e-10); s.forward()
        assert s.num_failed() == 0
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        t0 = time.perf_counter(); reps = 3
        k = np.zeros(3)
        for _ in range(reps):
            s.backward_async(1e-10); s.forward_async()
            if s.kernel_name != "generic":
                o = (C.c_double * 3)(); s._check(s._L.gar_hip_last_kernel_ms(s.handle, o)); k += np.array(list(o))
        s.sync(); dt = (time.perf_counter() - t0) / reps
        bwd = k[0] / reps
        frac = 8 * (knot + fac) * N * batch / (bwd * 1e-3) / 8e12 if bwd > 0 else float("nan")
        print(f"{s.kernel_name:12s} batch {batch:5d}: {batch / dt:9.0f} sweeps/s  ({dt * 1e3:8.2f} ms / sweep of the batch;"
              f" backward kernel {bwd:8.2f} ms = {frac:.3f} of 8 TB/s on the (56,22) algorithmic bytes)")
        s.close()

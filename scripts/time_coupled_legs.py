"""Leg mode on the reference's benchmark shape (36, 12, nc = 32), N = 256, ONE problem, with D = 0 (the fold onto the
wave-leg kernels, csrc/gar_fold.hpp) and with a random D on every knot (coupled constraints: round 6 the constrained
segment legs of csrc/gar_cstr_seg.hpp; GAR_HIP_CSTR_SEG_LEGS=0: the any-dimension leg kernels, as before): ms per backward
+ forward sweep beside the serial chain, and the kernels' own times from the library's events."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
import parity_cases as pc
nx, nu, nc, N, mu = 36, 12, 32, 256, 1e-8
batch = int(os.environ.get("BATCH", "1"))   # (BATCH > 1: the same problem in every slot)
for coupled in (False, True):
    prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
    if coupled:
        rng = np.random.default_rng(9)
        for k in prob.stages[:-1]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
    _, _, ref = pc.oracle_serial(prob, mu)
    sc = pc.scale_of(ref)
    for legs, seg in ((1, "1"), (6, "1"), (6, "0"), (32, "1"), (32, "0"), (64, "1")):
        if not coupled and seg == "0":
            continue
        os.environ["GAR_HIP_CSTR_SEG_LEGS"] = seg
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=batch, num_legs=legs)
        s.upload([prob] * batch)
        for _ in range(3):
            s.backward_async(mu); s.forward_async()
        s.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            s.backward_async(mu); s.forward_async()
        s.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        err = max(pc.maxdiff(a, b) for a, b in zip(s.solution(0)[:2], ref[:2])) / max(1.0, max(float(np.abs(v).max()) for v in ref[0]))
        print(f"batch {batch} D {'random' if coupled else '= 0   '} legs {legs:3d} {s.kernel_name:42s} {ms:8.3f} ms per sweep   |x, u - serial oracle| {err:.1e}", flush=True)
        s.close()

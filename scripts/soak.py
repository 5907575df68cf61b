"""Randomised parity soak on the GPU: many (shape, horizon, legs, generator, mu) draws through the
Python mirror against the CPU oracle.  Prints one line per failure and a summary."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aligator_amd import synth
import parity_cases as pc

rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "2026")))
budget = float(os.environ.get("SOAK_SECONDS", "150"))
shapes = [(36, 12, 0), (32, 12, 0), (16, 8, 0), (12, 8, 0), (12, 4, 0), (8, 4, 0), (12, 6, 0), (8, 3, 0),
          (36, 12, 32), (16, 8, 8), (8, 4, 4), (6, 3, 2), (5, 2, 0),
          (30, 10, 0), (13, 5, 0), (10, 3, 0), (7, 2, 0), (33, 11, 0), (40, 9, 0), (20, 14, 0),  # padded
          (56, 22, 0), (56, 24, 0), (50, 20, 0)]  # the wide family (two waves per problem / one wave)
if os.environ.get("SOAK_FOCUS") == "constrained":   # the three-kernel chain of constrained sweeps (gar_wave.hpp)
    shapes = [sh for sh in shapes if sh[2] > 0]
dense_share = float(os.environ.get("SOAK_DENSE", "0.25"))
t0, n, fails, kinds, why = time.time(), 0, 0, {}, {}
while time.time() - t0 < budget:
    nx, nu, nc = shapes[rng.integers(len(shapes))]
    horz = int(rng.integers(3, 70))
    mode = "F" if (rng.random() < 0.3 and nc == 0) else "W"      # the reference's generator, every shape
    # the THROUGHPUT kernel (one wave per problem; what the bench line runs: the library picks it for
    # batch > #CUs) on half of the serial draws, the latency kernel (one workgroup per problem) else
    os.environ["GAR_HIP_BACKWARD"] = ("wave", "wg4", "pair")[int(rng.integers(3))] if nx <= 36 else "wave"
    os.environ["GAR_HIP_WIDE"] = "pair" if rng.random() < 0.7 else "single"
    # (constrained problems below mu ~ 1e-10 are conditioned like 1/mu: the oracle and the kernels then
    # differ by cond * eps > 1e-6 from each other on EVERY kernel family, generic included)
    mu = 10.0 ** rng.uniform(-12 if nc == 0 else -10, -5)
    legs = 1 if (nx > 36 or rng.random() < (0.8 if nc > 0 else 0.4)) else int(rng.integers(2, max(3, min(9, horz // 2))))
    seed = int(rng.integers(1 << 30))
    prob = synth.generate_lq_problem(np.random.default_rng(seed), rng.standard_normal(nx), horz, nx, nu, nc=nc, mode=mode)
    if nc > 0:
        # D = 0 everywhere (the reference's generator) / on every knot / on a random subset: the sweep then
        # moves along the chain decoupled stage -> coupled stage (-> LDS Bunch-Kaufman where a knot's R, S are
        # scaled down so that the reduced KKT matrix pivots); a dense C half of the time
        what = rng.random()
        for k in prob.stages[:-1]:
            if what > 0.35 and (what > 0.7 or rng.random() < 0.3):
                k.D[...] = rng.uniform(-1, 1, k.D.shape)
                if rng.random() < 0.15:
                    k.R[...] *= 1e-3
                    k.S[...] *= 1e-3
        if rng.random() < 0.5:
            for k in prob.stages:
                k.C[...] = rng.uniform(-1, 1, k.C.shape)
    tol = pc.TOL[mode] if nc == 0 else 1e-6
    try:
        if rng.random() < dense_share:      # RiccatiSolverDense (csrc/gar_dense.hpp) against its own oracle
            # (constrained: gains and multipliers are compared stage by stage at 1e-5; two Bunch-Kaufman
            # implementations of the same 116x116 system differ by more than that (cond * eps)
            # below mu ~ 1e-9 -- the oracle and LAPACK differ by as much -- so mu >= 1e-8 here)
            if nc > 0:
                mu = max(mu, 1e-8)
            pc.check_dense(prob, mu, tol if nc == 0 else 1e-5)
            name, legs = "dense", 1
        elif legs == 1:
            s, _, _ = pc.check_serial(prob, mu, tol, factors=(nc == 0 or mu > 1e-9))
            name = s.kernel_name
        else:
            # (constrained problems in leg mode: the condensed leg-boundary system inherits the 1/mu
            # conditioning twice -- below mu ~ 5e-9 the leg-parallel and the serial solutions of the SAME
            # oracle differ by more than 1e-6 -- so mu >= 1e-8 here)
            if nc > 0:
                mu = max(mu, 1e-8)
            par = pc.check_parallel(prob, mu, legs, max(tol, 1e-8))
            name = par._impl.kernel_name
        kinds[name] = kinds.get(name, 0) + 1
    except Exception as e:
        fails += 1
        why[f"{type(e).__name__}: {str(e)[:60]}"] = why.get(f"{type(e).__name__}: {str(e)[:60]}", 0) + 1
        print(f"FAIL nx={nx} nu={nu} nc={nc} N={horz} legs={legs} mode={mode} mu={mu:.1e} seed={seed}: {type(e).__name__} {str(e)[:120]}")
        traceback.print_exc(limit=2)
    n += 1
print(f"soak: {n} problems, {fails} failures, kernels {kinds}")
for k, v in sorted(why.items(), key=lambda kv: -kv[1]):
    print(f"  {v:5d} x {k}")

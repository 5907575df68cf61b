"""Randomised parity soak on the GPU: many (shape, horizon, legs, generator, mu) draws through the
Python mirror against the CPU oracle.  Prints one line per failure and a summary.

The draw sequence depends on (SOAK_SEED, SOAK_FOCUS, SOAK_DENSE) only -- never on timing or on what the
checks found -- so `draws()` replays a logged failure exactly (scripts/soak_replay.py finds a draw by the
inner seed a FAIL line prints and saves the problem as a fixture)."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from soak_draws import SHAPES, draws  # noqa: E402,F401  (tests/soak_draws.py)


def main():
    import parity_cases as pc
    budget = float(os.environ.get("SOAK_SECONDS", "150"))
    dense_share = float(os.environ.get("SOAK_DENSE", "0.25"))
    t0, n, fails, kinds, why = time.time(), 0, 0, {}, {}
    for d in draws(os.environ.get("SOAK_SEED", "2026"), os.environ.get("SOAK_FOCUS"), version=int(os.environ.get("SOAK_VERSION", "2"))):
        if time.time() - t0 >= budget:
            break
        nx, nu, nc, horz, mode, mu, legs, seed, prob = (d[k] for k in ("nx", "nu", "nc", "horz", "mode", "mu", "legs", "seed", "prob"))
        os.environ["GAR_HIP_BACKWARD"] = d["backward"]
        os.environ["GAR_HIP_WIDE"] = d["wide"]
        tol = pc.TOL[mode] if nc == 0 else 1e-6
        try:
            if d["dense_draw"] < dense_share:      # RiccatiSolverDense (csrc/gar_dense.hpp) against its own oracle
                # (constrained: gains and multipliers are compared stage by stage at 1e-5; two Bunch-Kaufman
                # implementations of the same 116x116 system differ by more than that (cond * eps)
                # below mu ~ 1e-9 -- the oracle and LAPACK differ by as much -- so mu >= 1e-8 here)
                if nc > 0:
                    mu = max(mu, 1e-8)
                pc.check_dense(prob, mu, tol if nc == 0 else 1e-5)
                name, legs = "dense", 1
            elif legs == 1:
                s, _, _ = pc.check_serial(prob, mu, tol, factors=(nc == 0 or mu > 1e-9), conditioned=(nc > 0))
                name = s.kernel_name
            else:
                # constrained problems in leg mode: the condensed leg-boundary system inherits the 1/mu
                # conditioning of the value function twice.  The bound is part of the check itself
                # (parity_cases.leg_mode_tolerance: relative to what the oracle's own leg-parallel and serial
                # solutions differ by and to the LAPACK dense-KKT arbitration), not a floor on mu here.
                par = pc.check_parallel(prob, mu, legs, max(tol, 1e-8), conditioned=(nc > 0))
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


if __name__ == "__main__":
    main()

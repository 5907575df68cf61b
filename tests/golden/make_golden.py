#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/*.npz (test infrastructure).

The reference ships NO golden vectors for the gar path and cannot be executed in
this image (it needs Eigen 3.4), so these fixtures are NOT
outputs of the reference binary.  They are reference-*defined* instead: each one
is an LQ problem shaped like one of the reference's own tests
(tests/gar/riccati.cpp, tests/gar/parallel.cpp, bench/gar-riccati.cpp) together
with the solution of its global dense KKT system, assembled exactly like the
reference's test helper (tests/gar/test_util.hpp:92-165) and solved with LAPACK
plus iterative refinement in extended precision.  That solve shares no code with
oracle/gar_oracle.c nor with the HIP kernels: the oracle AND the HIP path are
both checked against these files (tests/test_golden.py).

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from aligator_amd import synth                     # noqa: E402
from aligator_amd.lqr import BLOCK_NAMES          # noqa: E402
from oracle import dense_kkt                      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def refined_dense_solve(problem, mueq, theta=None):
    """-K^{-1} rhs with three steps of iterative refinement (residual in long double)."""
    mat, rhs = dense_kkt.lqr_dense_matrix(problem, mueq)
    if theta is not None:  # theta enters the stationarity rows (gar/utils.hxx:118-166)
        idx = problem.nc0
        for t, k in enumerate(problem.stages):
            rhs[idx:idx + k.nx] += k.Gx @ theta
            rhs[idx + k.nx:idx + k.nx + k.nu] += k.Gu @ theta
            rhs[idx + k.nx + k.nu:idx + k.nx + k.nu + k.nc] += k.Gv @ theta
            idx += k.nx + k.nu + k.nc + (k.nx2 if t < problem.horizon else 0)
    import scipy.linalg as sla
    lu = sla.lu_factor(mat)
    sol = -sla.lu_solve(lu, rhs)
    matl, rhsl = mat.astype(np.longdouble), rhs.astype(np.longdouble)
    for _ in range(3):
        res = (-rhsl - matl @ sol.astype(np.longdouble)).astype(np.float64)
        sol = sol + sla.lu_solve(lu, res)
    return dense_kkt.dense_solution_to_traj(problem, sol)


def pack(problem):
    out = {"dims": np.array([k.dims for k in problem.stages], dtype=np.int32),
           "G0": problem.G0.copy(), "g0": problem.g0.copy()}
    for t, k in enumerate(problem.stages):
        for name in BLOCK_NAMES:
            a = getattr(k, name)
            if a.size:
                out[f"k{t}_{name}"] = np.array(a, dtype=np.float64)
    return out


def cases():
    rng = np.random.default_rng(20240607)
    yield "short_horz_8", synth.short_horizon_problem(8), 1e-14, None            # riccati.cpp:26-85
    yield "one_knot", synth.generate_lq_problem(1, np.zeros(2), 0, 2, 2), 1e-13, None  # :87-105
    yield "random_W_nx6", synth.generate_lq_problem(11, np.zeros(6), 12, 6, 3, mode="W"), 1e-12, None
    yield "random_F_nx6", synth.generate_lq_problem(12, np.zeros(6), 12, 6, 3, mode="F"), 1e-12, None
    yield "constrained_nx6_nc4", synth.generate_lq_problem(rng, rng.standard_normal(6), 5, 6, 3, nc=4, mode="W"), 1e-9, None
    p = synth.generate_lq_problem(rng, rng.standard_normal(5), 10, 5, 2, nth=1, mode="W")  # :157-192
    yield "parametric_nx5", p, 1e-12, rng.uniform(-1, 1, 1)
    yield "north_star_shape_N6", synth.generate_lq_problem(7, np.zeros(36), 6, 36, 12, mode="W"), 1e-14, None
    yield "north_star_shape_F_N4", synth.generate_lq_problem(8, np.zeros(36), 4, 36, 12, mode="F"), 1e-14, None
    yield "parallel_shape_nx8_N17", synth.generate_lq_problem(rng, np.zeros(8), 17, 8, 3, mode="W"), 1e-9, None  # parallel.cpp:185-245
    yield "mfma_shape_nx32_N5", synth.generate_lq_problem(9, np.zeros(32), 5, 32, 12, mode="W"), 1e-12, None


def main():
    for name, prob, mueq, theta in cases():
        xs, us, vs, lbdas = refined_dense_solve(prob, mueq, theta)
        data = pack(prob)
        data["mueq"] = np.float64(mueq)
        data["theta"] = np.zeros(0) if theta is None else np.asarray(theta)
        for nm, part in (("xs", xs), ("us", us), ("vs", vs), ("lbdas", lbdas)):
            data[nm] = np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: N={prob.horizon} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()

"""Synthetic LQ-problem generators (numpy, host side).

Restates the reference's test fixtures so that the parity tests read like the
reference's own (tests/gar/test_util.cpp:14-76, tests/test_util.hpp:4-23):

* generator "F" (faithful): ``[Q S; S^T R] = W / max(nx, nu)`` with
  ``W = G G^T``, ``G in R^{(nx+nu) x (nx+nu+1)} ~ N(0,1)``; optional singular
  ``Q = G2 G2^T`` with ``0.8 (nx+nu)`` columns; ``R_ii *= 1 + 1e-6``;
  ``q, r, A, B, d ~ U[-1, 1]``; ``f ~ N(0,1)``; ``C = I``; last knot ``nu = 0``.
* generator "W" (well-conditioned, ours):
  same but ``A = I + 0.1 U[-1,1]``, ``B = 0.5 U[-1,1]``.

Unlike the reference (whose Wishart sampler re-seeds a default mt19937 on
every call, and whose ``generateKnot`` takes the rng by value so every knot
is identical) every knot here gets fresh draws from ``default_rng(seed)``.
"""
from __future__ import annotations

import numpy as np

from .lqr import LqrKnot, LqrProblem


def sample_wishart(rng: np.random.Generator, n: int, p: int) -> np.ndarray:
    """tests/test_util.hpp:17-23 (root * root^T, root n x p standard normal)."""
    root = rng.standard_normal((n, p))
    return root @ root.T


def generate_knot(rng, nx, nu, nc=0, nth=0, singular=False, nx2=None, mode="F") -> LqrKnot:
    """tests/gar/test_util.cpp:14-56."""
    k = LqrKnot(nx, nu, nc, nx2, nth)
    nx2 = k.nx2
    qsr = sample_wishart(rng, nx + nu, nx + nu + 1) / max(nx, nu)
    k.Q[...] = qsr[:nx, :nx]
    k.S[...] = qsr[:nx, nx:]
    if singular:
        dof = int(0.8 * (nx + nu))
        k.Q[...] = sample_wishart(rng, nx, dof)
    k.R[...] = qsr[nx:, nx:]
    k.R[np.diag_indices(nu)] *= 1.0 + 1e-6
    k.q[...] = rng.uniform(-1, 1, nx)
    k.r[...] = rng.uniform(-1, 1, nu)
    if mode == "F":
        k.A[...] = rng.uniform(-1, 1, (nx2, nx))
        k.B[...] = rng.uniform(-1, 1, (nx2, nu))
    elif mode == "W":
        k.A[...] = np.eye(nx2, nx) + 0.1 * rng.uniform(-1, 1, (nx2, nx))
        k.B[...] = 0.5 * rng.uniform(-1, 1, (nx2, nu))
    else:
        raise ValueError(mode)
    k.f[...] = rng.standard_normal(nx2)
    if nc > 0:
        k.C[...] = np.eye(nc, nx)
        k.d[...] = rng.uniform(-1, 1, nc)
    if nth > 0:
        k.Gx[...] = rng.standard_normal((nx, nth))
        k.Gu[...] = rng.standard_normal((nu, nth))
        k.Gth[...] = sample_wishart(rng, nth, nth + 2)
        k.gamma[...] = rng.standard_normal(nth)
    return k


def generate_lq_problem(seed, x0, horz, nx, nu, nth=0, nc=0, singular=True,
                        mode="F") -> LqrProblem:
    """tests/gar/test_util.cpp:58-76: ``G0 = -I``, ``g0 = x0``; the terminal
    knot has ``nu = 0`` and a non-singular Q."""
    rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)
    x0 = np.asarray(x0, dtype=np.float64)
    knots = [generate_knot(rng, nx, nu, nc, nth, singular, mode=mode)
             for _ in range(horz)]
    knots.append(generate_knot(rng, nx, 0, nc, nth, False, mode=mode))
    prob = LqrProblem(knots, nx)
    prob.g0[...] = x0
    prob.G0[...] = -np.eye(nx)
    return prob


def short_horizon_problem(horz: int, seed: int = 0) -> LqrProblem:
    """tests/gar/riccati.cpp:26-59 (riccati_short_horz_pb): nx = nu = 2, one
    stage (t = 4) carries an nc = 2 constraint with D = I, d = 0.1."""
    rng = np.random.default_rng(seed)
    nx = nu = 2
    x0 = np.ones(nx)
    x1 = -np.ones(nx)

    def init_knot(nc):
        k = LqrKnot(nx, nu, nc)
        k.A[...] = np.array([[0.1, 0.0], [-0.1, 0.01]])
        k.B[...] = rng.uniform(-1, 1, (nx, nu))
        k.f[...] = rng.uniform(-1, 1, nx)
        k.Q[...] = 0.01 * np.eye(nx)
        k.R[...] = 0.1 * np.eye(nu)
        return k

    base = init_knot(0)
    knot1 = base.copy()
    knot1.Q[...] = np.eye(nx)
    knot1.q[...] = -x1
    knots = [base.copy() for _ in range(horz + 1)]
    knots[4] = init_knot(nu)
    knots[4].D[...] = np.eye(nu)
    knots[4].d[...] = 0.1
    knots[horz] = knot1
    prob = LqrProblem(knots, nx)
    prob.g0[...] = -x0
    prob.G0[...] = np.eye(nx)
    return prob


def randomly_modify_problem(rng, prob: LqrProblem) -> None:
    """tests/gar/parallel.cpp:173-183."""
    N = prob.horizon
    for i in (0, N // 3, N // 2, N // 2 + 1, N // 2 + 2, N):
        kn = prob.stages[i]
        kn.A += 0.1 * rng.standard_normal(kn.A.shape)
        kn.B += 0.1 * rng.standard_normal(kn.B.shape)
        kn.q += 0.1 * rng.standard_normal(kn.q.shape)

"""CPU: the stage-dense oracle (oracle/dense_riccati.py, restating gar/dense-kernel.hpp and
gar/dense-riccati.hxx) pinned against an independent LAPACK solve of the global dense KKT system
(oracle/dense_kkt.py) and against the Riccati oracle, at the reference's bar for this solver
(tests/gar/riccati.cpp:141-155: KKT error <= 1e-8 at nx=36, nu=12)."""
import numpy as np
import pytest

from aligator_amd import synth
from aligator_amd.gar import lqrComputeKktError, lqrInitializeSolution
from oracle import dense_kkt
from oracle.dense_riccati import RiccatiSolverDense


def _maxdiff(a, b):
    return max((np.max(np.abs(x - y)) if x.size else 0.0) for x, y in zip(a, b))


def _solve(prob, mueq, theta=None, **kw):
    s = RiccatiSolverDense(prob, **kw)
    assert s.backward(mueq)
    sol = lqrInitializeSolution(prob)
    assert s.forward(*sol, theta)
    return s, sol


@pytest.mark.parametrize("mode,horz", [("W", 100), ("F", 40)])
def test_dense_random_large_problem(mode, horz):             # riccati.cpp:107-155
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(42, np.zeros(nx), horz, nx, nu, mode=mode)
    s, sol = _solve(prob, 1e-14)
    assert max(lqrComputeKktError(prob, *sol)) <= 1e-8        # :154
    ref = dense_kkt.dense_solve(prob, 1e-14)
    scale = max(np.abs(np.concatenate(ref[3])).max(), 1.0)
    for a, b in zip(sol, ref):
        assert _maxdiff(a, b) <= (1e-9 if mode == "W" else 1e-6) * scale
    assert s.getFeedback(0).shape == (nu + 2 * nx, nx) and s.getFeedforward(0).shape == (nu + 2 * nx,)


def test_dense_constrained_and_terminal_block():
    """Constraints on every knot, the terminal one included.  The leading-block terminal solve
    (what the HIP kernel does) satisfies the KKT conditions; the reference's literal full-matrix
    factorisation only coincides with it when the terminal knot has nu = nc = 0."""
    nx, nu, nc, mu = 8, 4, 3, 1e-6
    prob = synth.generate_lq_problem(7, np.ones(nx), 12, nx, nu, nc=nc, mode="W")
    _, sol = _solve(prob, mu)
    assert max(lqrComputeKktError(prob, *sol, mueq=mu)) <= 1e-8
    ref = dense_kkt.dense_solve(prob, mu)
    scale = max(max(np.abs(np.concatenate(r)).max() for r in ref if len(r)), 1.0)
    for a, b in zip(sol, ref):
        assert _maxdiff(a, b) <= 1e-9 * scale
    free = synth.generate_lq_problem(7, np.ones(nx), 12, nx, nu, mode="W")     # nu = nc = 0 at the end
    _, a = _solve(free, mu)
    _, b = _solve(free, mu, terminal_leading_block=False)
    for x, y in zip(a, b):
        assert _maxdiff(x, y) == 0.0


def test_dense_parametric():                                  # riccati.cpp:157-192, on the dense solver
    rng = np.random.default_rng(9)
    nx, nu, nth = 10, 4, 2
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 30, nx, nu, nth=nth, mode="W")
    theta = rng.uniform(-1, 1, nth)
    s, sol = _solve(prob, 1e-12, theta)
    assert max(lqrComputeKktError(prob, *sol, mueq=1e-12, theta=theta)) <= 1e-9
    for arr in (s.kkt0_ff, s.kkt0_fth, s.thGrad, s.thHess, s.pt[0], s.Ptt[30]):
        assert np.isfinite(arr).all()
    assert np.allclose(s.thHess, s.thHess.T, atol=1e-9)       # Hessian of the optimal value in theta

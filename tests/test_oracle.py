"""Pins for the CPU oracle (oracle/gar_oracle.c).

The reference ships no golden vectors for the gar path,
so the oracle is pinned by (1) the reference's own test assertions, restated
below with the same thresholds (tests/gar/riccati.cpp, tests/gar/parallel.cpp,
tests/block-matrix.cpp), and (2) an independent LAPACK solve of the global
dense KKT system (oracle/dense_kkt.py).
"""
import numpy as np
import pytest

from aligator_amd import synth
from aligator_amd.lqr import lqrComputeKktError, lqrInitializeSolution
from oracle import dense_kkt


def _to_oracle(ora, prob):
    return ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)


def _solve_serial(ora, prob, mueq, theta=None):
    op = _to_oracle(ora, prob)
    solver = ora.ProximalRiccatiSolver(op)
    assert solver.backward(mueq)
    xs, us, vs, lbdas = lqrInitializeSolution(prob)
    assert solver.forward(xs, us, vs, lbdas, theta)
    return op, solver, (xs, us, vs, lbdas)


def _maxdiff(a, b):
    return max((np.max(np.abs(x - y)) if x.size else 0.0) for x, y in zip(a, b))


# --- BunchKaufman (core/bunchkaufman.hpp) -----------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 5, 12, 31, 32, 33, 44, 72, 100])
def test_bunch_kaufman_random_indefinite(oracle, n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    a = a + a.T
    bk = oracle.BunchKaufman(a)
    assert bk.info == 0
    b = rng.standard_normal((n, 3))
    x = bk.solve(b)
    assert np.allclose(a @ x, b, atol=1e-9 * max(1.0, np.abs(x).max()))
    ref = np.linalg.solve(a, b)
    assert np.allclose(x, ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())


@pytest.mark.parametrize("n,nc", [(4, 2), (12, 32), (22, 20)])
def test_bunch_kaufman_kkt_2x2_pivots(oracle, n, nc):
    """[R D^T; D -mu I] with tiny mu forces 2x2 pivots (riccati-kernel.hxx:232-238)."""
    rng = np.random.default_rng(7)
    r = synth.sample_wishart(rng, n, n + 1)
    d = rng.standard_normal((nc, n))
    m = np.block([[r, d.T], [d, -1e-11 * np.eye(nc)]])
    bk = oracle.BunchKaufman(m)
    assert bk.info == 0
    b = rng.standard_normal(n + nc)
    x = bk.solve(b)
    assert np.linalg.norm(m @ x - b, np.inf) <= 1e-9 * max(1.0, np.abs(x).max())


def test_bunch_kaufman_zero_diagonal_and_failure(oracle):
    a = np.array([[0.0, 1.0], [1.0, 0.0]])
    bk = oracle.BunchKaufman(a)
    assert bk.info == 0 and (bk.pivots < 0).all()  # one 2x2 pivot
    assert np.allclose(bk.solve(np.array([2.0, 3.0])), [3.0, 2.0])
    assert oracle.BunchKaufman(np.zeros((3, 3))).info == 1  # NumericalIssue (:58-59)


def test_bunch_kaufman_reads_lower_triangle_only(oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((9, 9))
    a = a + a.T
    junk = np.tril(a) + np.triu(rng.standard_normal((9, 9)), 1)
    b = rng.standard_normal(9)
    assert np.array_equal(oracle.BunchKaufman(a).solve(b),
                          oracle.BunchKaufman(junk).solve(b))


# --- block tridiagonal (tests/block-matrix.cpp:58-120) ----------------------
@pytest.mark.parametrize("down", [False, True])
def test_block_tridiag_solve(oracle, down):
    rng = np.random.default_rng(11)
    N, nx = 6, 2
    B = rng.standard_normal((nx, nx))
    diag = [synth.sample_wishart(rng, nx, nx + 1) for _ in range(N + 1)]
    sup = [B.copy() for _ in range(N)]
    sub = [B.T.copy() for _ in range(N)]
    rhs = [np.ones(nx) for _ in range(N + 1)]
    ok, sol = oracle.block_tridiag_solve(sub, diag, sup, rhs, down=down)
    assert ok
    n = (N + 1) * nx
    dense = np.zeros((n, n))
    for i in range(N + 1):
        dense[i * nx:(i + 1) * nx, i * nx:(i + 1) * nx] = diag[i]
        if i < N:
            dense[i * nx:(i + 1) * nx, (i + 1) * nx:(i + 2) * nx] = sup[i]
            dense[(i + 1) * nx:(i + 2) * nx, i * nx:(i + 1) * nx] = sub[i]
    ref = oracle.BunchKaufman(dense).solve(np.ones(n))
    assert np.allclose(np.concatenate(sol), ref, rtol=1e-12, atol=1e-12)


# --- tests/gar/riccati.cpp --------------------------------------------------
@pytest.mark.parametrize("horz", [4, 8, 16])
def test_riccati_short_horz_pb(oracle, horz):
    mueq = 1e-14
    prob = synth.short_horizon_problem(horz)
    op, solver, sol = _solve_serial(oracle, prob, mueq)
    assert max(oracle.lqr_kkt_error(op, *sol)) <= 1e-9   # riccati.cpp:84
    assert max(lqrComputeKktError(prob, *sol)) <= 1e-9    # host restatement agrees
    ref = dense_kkt.dense_solve(prob, mueq)
    for a, b in zip(sol, ref):
        assert _maxdiff(a, b) <= 1e-9


def test_riccati_one_knot_prob(oracle):
    prob = synth.generate_lq_problem(1, np.zeros(2), 0, 2, 2)
    op, solver, (xs, us, vs, lbdas) = _solve_serial(oracle, prob, 1e-13)
    assert len(xs) == 1 and len(us) == 0 and len(lbdas) == 1
    assert max(oracle.lqr_kkt_error(op, xs, us, vs, lbdas)) <= 1e-10  # :104


@pytest.mark.parametrize("horz", [20, 100])
@pytest.mark.parametrize("mode", ["F", "W"])
def test_riccati_random_large_problem(oracle, horz, mode):
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(42, np.zeros(nx), horz, nx, nu, mode=mode)
    op, solver, sol = _solve_serial(oracle, prob, 1e-14)
    err = max(oracle.lqr_kkt_error(op, *sol))
    assert err <= (1e-6 if mode == "F" else 1e-9)           # riccati.cpp:138
    ref = dense_kkt.dense_solve(prob, 1e-14)
    scale = max(np.abs(np.concatenate(ref[3])).max(), 1.0)
    tol = 1e-6 if mode == "F" else 1e-9
    for a, b in zip(sol, ref):
        assert _maxdiff(a, b) <= tol * scale


def test_riccati_constrained_bench_shape(oracle):
    """bench/gar-riccati.cpp:19-22 shape (nc=32, mu=1e-11); the reference never
    checks residuals here; SURVEY Appendix B: use relative metrics."""
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 16, nx, nu, nc=nc, mode="W")
    mueq = 1e-11
    op, solver, sol = _solve_serial(oracle, prob, mueq)
    # multipliers are O(1/mu) ~ 4e11 here, so residuals and the LAPACK
    # cross-check are only meaningful relative to that scale (kappa ~ 1e11).
    vnorm = max(np.abs(solver.datas(t).Vxx).max() for t in range(prob.horizon + 1))
    dyn, cst, dual = oracle.lqr_kkt_error(op, *sol, mueq=mueq)
    assert dyn <= 1e-9 and cst <= 1e-9 and dual <= 1e-13 * vnorm
    ref = dense_kkt.dense_solve(prob, mueq)
    for a, b in zip(sol[:2], ref[:2]):                      # xs, us
        assert _maxdiff(a, b) <= 5e-3
    assert _maxdiff(sol[2], ref[2]) <= 1e-3 * vnorm          # vs = (Cx+d)/mu


def test_riccati_parametric(oracle):
    rng = np.random.default_rng(9)
    nx, nu, horz, nth = 10, 4, 100, 1
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nth=nth)
    theta = rng.uniform(-1, 1, nth)
    op, solver, sol = _solve_serial(oracle, prob, 1e-12, theta)
    assert max(oracle.lqr_kkt_error(op, *sol, theta=theta)) <= 1e-9  # :178
    assert max(lqrComputeKktError(prob, *sol, theta=theta)) <= 1e-9
    for arr in (solver.kkt0_ff, solver.kkt0_fth, solver.thGrad, solver.thHess,
                solver.datas(0).vt, solver.datas(0).Vxt, solver.datas(0).Vtt,
                solver.datas(horz).vt, solver.datas(horz).Vxt, solver.datas(horz).Vtt):
        assert np.isfinite(arr).all()                                  # :181-191


# --- tests/gar/parallel.cpp -------------------------------------------------
def _split_problem_in_two(prob, t0):
    """tests/gar/parallel.cpp:20-59."""
    from aligator_amd.lqr import LqrProblem
    nx_t0 = prob.stages[t0].nx
    p1 = LqrProblem([k.copy() for k in prob.stages[:t0]], prob.nc0)
    p1.G0[...] = prob.G0
    p1.g0[...] = prob.g0
    last = prob.stages[t0 - 1]
    p1.addParameterization(nx_t0)
    p1.stages[-1].Gx[...] = last.A.T
    p1.stages[-1].Gu[...] = last.B.T
    p1.stages[-1].gamma[...] = last.f
    p2 = LqrProblem([k.copy() for k in prob.stages[t0:]], 0)
    p2.addParameterization(nx_t0)
    p2.stages[0].Gx[...] = -np.eye(nx_t0)
    return p1, p2


def test_parallel_manual(oracle):
    rng = np.random.default_rng(13)
    nx = nu = 2
    horizon, mueq, EPS = 16, 1e-14, 1e-9
    prob = synth.generate_lq_problem(rng, rng.uniform(-1, 1, nx), horizon, nx, nu)
    _, _, full = _solve_serial(oracle, prob, mueq)
    t0 = horizon // 2
    p1, p2 = _split_problem_in_two(prob, t0)
    assert p1.horizon + p2.horizon + 1 == horizon
    o1, o2 = _to_oracle(oracle, p1), _to_oracle(oracle, p2)
    s1, s2 = oracle.ProximalRiccatiSolver(o1), oracle.ProximalRiccatiSolver(o2)
    s1.backward(mueq)
    s2.backward(mueq)
    th = np.linalg.solve(s1.thHess + s2.thHess, -(s1.thGrad + s2.thGrad))
    sol1, sol2 = lqrInitializeSolution(p1), lqrInitializeSolution(p2)
    s1.forward(*sol1, th)
    s2.forward(*sol2, th)
    assert max(oracle.lqr_kkt_error(o1, *sol1, theta=th)) <= EPS   # :131
    assert max(oracle.lqr_kkt_error(o2, *sol2, theta=th)) <= EPS   # :132
    xs_m = sol1[0] + sol2[0]
    us_m = sol1[1] + sol2[1]
    lb_m = sol1[3] + [th] + sol2[3][1:]
    assert _maxdiff(xs_m, full[0]) <= 1e-8
    assert _maxdiff(us_m, full[1]) <= 1e-8
    assert _maxdiff(lb_m, full[3]) <= 1e-8


@pytest.mark.parametrize("nthreads", [2, 3, 6])
def test_parallel_solver_class(oracle, nthreads):
    rng = np.random.default_rng(17)
    nx, nu, horizon, TOL, mueq = 32, 12, 50, 1e-7, 1e-9
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horizon, nx, nu)
    _, _, ref = _solve_serial(oracle, prob, mueq)
    op = _to_oracle(oracle, prob)
    par = oracle.ParallelRiccatiSolver(op, nthreads)
    par.maxRefinementSteps = 10
    sol = lqrInitializeSolution(prob)
    assert par.backward(mueq)
    assert par.forward(*sol)
    assert max(oracle.lqr_kkt_error(op, *sol, mueq=mueq)) <= TOL   # :221
    assert _maxdiff(sol[0], ref[0]) <= TOL                          # :234
    assert _maxdiff(sol[3], ref[3]) <= TOL                          # :235
    for _ in range(3):                                              # :238-244
        for i in (0, horizon // 3, horizon // 2, horizon // 2 + 1, horizon // 2 + 2, horizon):
            kn = op.knot(i)
            kn.A[...] += 0.1 * rng.standard_normal(kn.A.shape)
            kn.B[...] += 0.1 * rng.standard_normal(kn.B.shape)
            kn.q[...] += 0.1 * rng.standard_normal(kn.q.shape)
        par.backward(mueq)
        par.forward(*sol)
        assert max(oracle.lqr_kkt_error(op, *sol, mueq=mueq)) <= TOL


def test_parallel_rejects_single_thread(oracle):
    prob = synth.generate_lq_problem(1, np.zeros(2), 4, 2, 2)
    with pytest.raises(RuntimeError):
        oracle.ParallelRiccatiSolver(_to_oracle(oracle, prob), 1)   # parallel-solver.hxx:42-46


def test_get_work_partition(oracle):
    for horz, nt in [(50, 6), (256, 8), (7, 2), (2047, 8)]:
        cover = []
        for i in range(nt):
            b, e = oracle.get_work(horz, i, nt)
            cover += list(range(b, e))
        assert cover == list(range(horz + 1))


@pytest.mark.parametrize("soak_seed,inner_seed,focus,legs", [(101, 353783436, None, 2), (2026, 755480262, "constrained", 8)])
def test_soak_failures_are_conditioning_on_the_cpu_side(oracle, soak_seed, inner_seed, focus, legs):
    ora = oracle
    """The CPU half of tests/test_gpu_parity.py::test_soak_failures_replayed_and_arbitrated: on the two
    round-2 soak draws (nc = 32, leg mode, mu = 1e-8) the oracle's serial and leg-parallel solutions and LAPACK
    on the dense KKT matrix agree to 1e-14 in x, u and differ by 1e-8 ... 3e-6 (relative to the O(1/mu)
    multipliers) in v, lambda -- the spread any fourth solver has to be judged against."""
    import parity_cases as pc
    from soak_draws import find_draw
    from aligator_amd.lqr import lqrInitializeSolution
    d = find_draw(soak_seed, inner_seed, focus)
    prob, mu = d["prob"], max(d["mu"], 1e-8)
    _, _, ref = pc.oracle_serial(prob, mu)
    bound, lap = pc.conditioning_bound(prob, mu, ref)
    opar = ora.ParallelRiccatiSolver(pc.to_oracle(prob), legs)
    opar.maxRefinementSteps = 10
    opar.backward(mu)
    osol = lqrInitializeSolution(prob)
    opar.forward(*osol)
    sc = pc.scale_of(ref)
    leg = [pc.maxdiff(a, b) / sc for a, b in zip(osol, ref)]
    assert sc > 1e9                                           # multipliers of order 1/mu
    assert max(bound[:2]) < 1e-13 and max(leg[:2]) < 1e-13    # x, u: well determined
    assert 1e-8 < max(bound[2:]) < 1e-5 and max(leg[2:]) < 1e-5

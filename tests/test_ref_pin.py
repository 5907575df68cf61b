"""Pins the CPU oracle (oracle/gar_oracle.c) against the REFERENCE'S OWN CODE, executed live: the reference's gar
sources compile unchanged over the minimal Eigen-API stand-in oracle/ref_shim into oracle/_ref/libgar_ref.so
(oracle/ref_build.sh; Eigen itself is absent from this image -- SURVEY.md section 8c asked for exactly this route).
What is compared: the in-tree Bunch-Kaufman (pivot sequences identical, factors and solves to rounding; unblocked
n <= 32 and blocked n > 32 paths), the serial solver (solution, every StageFactor block, kkt0, thGrad / thHess), the
leg-parallel solver (solution, factors, condensed solution, the collapsed K0 with the reference's literal
`subdiagonal[1]`), and both block-tridiagonal solves.  The shim evaluates eagerly, so sums inside a dot product may
associate differently from real Eigen's kernels -- as between any two BLAS; every decision (pivot rule, ordering of
operations, which triangle is read, what is symmetrised when) is the reference's own compiled code.
The committed vectors tests/golden/ref/*.npz carry the same outputs to the GPU box (tests/test_golden.py)."""
import numpy as np
import pytest

from aligator_amd import synth
from aligator_amd.lqr import lqrInitializeSolution
from oracle import ref
import parity_cases as pc

pytestmark = pytest.mark.skipif(not ref.available(), reason="neither /root/reference nor a prebuilt oracle/_ref")


def test_bunch_kaufman_pivots_identical_and_values_to_rounding(oracle):
    rng = np.random.default_rng(0)
    n2 = nsw = 0
    for trial in range(300):
        n = int(rng.integers(1, 90))               # > 32: the reference's blocked path (bunchkaufman.hpp:172-344)
        kind = trial % 4
        G = rng.standard_normal((n, n))
        if kind == 0:
            A = G + G.T                                              # indefinite: interchanges and 2x2 pivots
        elif kind == 1:
            A = G @ G.T + 1e-3 * np.eye(n)
        elif kind == 2:                                              # [H D^T; D -mu I]
            k = n // 2
            A = np.zeros((n, n))
            A[:k, :k] = G[:k, :k] @ G[:k, :k].T
            A[k:, :k] = rng.uniform(-1, 1, (n - k, k))
            A[:k, k:] = A[k:, :k].T
            A[k:, k:] = -1e-4 * np.eye(n - k)
        else:
            A = np.diag(rng.standard_normal(n)) + 1e-3 * (G + G.T)
        info, ldlt, sub, piv = ref.bk_compute(A)
        o = oracle.BunchKaufman(A)
        assert info == o.info
        assert np.array_equal(piv, o.pivots), (n, kind)
        n2 += int((piv < 0).sum())
        nsw += int(((piv >= 0) & (piv != np.arange(n))).sum())
        cond = np.linalg.cond(A)
        tol = 1e-13 * max(cond, 10.0)
        assert np.abs(ldlt - o.matrixLDLT).max() <= tol * max(1.0, np.abs(ldlt).max()), (n, kind)
        assert np.abs(sub - o.subdiag).max() <= tol * max(1.0, np.abs(sub).max())
        B = rng.standard_normal((n, 3))
        X, Xo = ref.bk_solve(A, B), o.solve(B)
        assert np.abs(X - Xo).max() <= tol * max(1.0, np.abs(Xo).max())
    assert n2 > 500 and nsw > 200                  # the comparison did exercise the pivoting


CASES = [
    ("short_horz", lambda: synth.short_horizon_problem(8), 1e-14, None, 1e-12),
    ("F_nx36_N20", lambda: synth.generate_lq_problem(42, np.zeros(36), 20, 36, 12, mode="F"), 1e-14, None, 1e-9),
    ("W_nx36_N100", lambda: synth.generate_lq_problem(42, np.zeros(36), 100, 36, 12, mode="W"), 1e-14, None, 1e-12),
    ("wide_56_22", lambda: synth.generate_lq_problem(4, np.zeros(56), 12, 56, 22, mode="W"), 1e-12, None, 1e-12),
]


def _parametric():
    rng = np.random.default_rng(9)
    return synth.generate_lq_problem(rng, rng.standard_normal(10), 30, 10, 4, nth=2), rng.uniform(-1, 1, 2)


def _constrained(nx, nu, nc, horz, seed, dense_d=True):
    rng = np.random.default_rng(seed)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    if dense_d:
        for k in prob.stages[:-1]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
    return prob


def _compare_serial(oracle, prob, mu, theta, tol):
    rs = ref.ProximalRiccatiSolver(ref.Problem(prob))
    assert rs.backward(mu)
    rsol = rs.forward(theta)
    _, osol, oref = pc.oracle_serial(prob, mu, theta)
    sc = pc.scale_of(oref)
    for a, b in zip(rsol, oref):
        assert pc.maxdiff(a, b) <= tol * sc
    for t in range(prob.horizon + 1):
        f, o = rs.datas(t), osol.datas(t)
        for nm in ("ff", "fb", "fth", "Vxx", "vx", "Vxt", "Vtt", "vt", "kktMat"):
            a, b = getattr(f, nm), getattr(o, nm)
            if a.size:
                assert a.shape == b.shape
                assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (t, nm)
        if prob.stages[t].nu > 0:
            assert np.array_equal(f.pivots, oracle.BunchKaufman(o.kktMat).pivots), t
    for a, b in zip(rs.initial(), (osol.kkt0_ff, osol.kkt0_fth, osol.thGrad, osol.thHess)):
        if a.size:
            assert np.abs(a - np.asarray(b).reshape(a.shape)).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name,make,mu,theta,tol", CASES, ids=[c[0] for c in CASES])
def test_serial_solver_oracle_equals_reference_code(oracle, name, make, mu, theta, tol):
    _compare_serial(oracle, make(), mu, theta, tol)


def test_serial_solver_parametric_and_constrained(oracle):
    prob, theta = _parametric()
    _compare_serial(oracle, prob, 1e-12, theta, 1e-11)
    _compare_serial(oracle, _constrained(6, 3, 4, 12, 5), 1e-8, None, 1e-11)
    _compare_serial(oracle, _constrained(36, 12, 32, 10, 6, dense_d=False), 1e-11, None, 1e-9)   # the reference's bench shape
    _compare_serial(oracle, _constrained(36, 12, 32, 6, 7, dense_d=True), 1e-8, None, 1e-9)     # n = 44 > 32: blocked BK per stage


@pytest.mark.parametrize("legs", [2, 3, 6])
def test_parallel_solver_oracle_equals_reference_code(oracle, legs):
    rng = np.random.default_rng(17)
    for prob, mu, tol in ((synth.generate_lq_problem(rng, np.zeros(32), 50, 32, 12), 1e-9, 1e-9),
                          (synth.generate_lq_problem(3, np.zeros(12), 128, 12, 6, mode="W"), 1e-9, 1e-12),
                          (_constrained(8, 4, 3, 17, 3, dense_d=False), 1e-7, 1e-9)):
        rpar = ref.ParallelRiccatiSolver(ref.Problem(prob), legs)
        rpar.set_refinement(1e-10, 10)
        assert rpar.backward(mu)
        rsol = rpar.forward()
        opar = oracle.ParallelRiccatiSolver(pc.to_oracle(prob), legs)
        opar.maxRefinementSteps = 10
        opar.backward(mu)
        osol = lqrInitializeSolution(prob)
        opar.forward(*osol)
        sc = pc.scale_of(osol)
        for a, b in zip(rsol, osol):
            assert pc.maxdiff(a, b) <= tol * sc
        for t in range(prob.horizon + 1):
            f, o = rpar.datas(t), opar.datas(t)
            for nm in ("ff", "fb", "fth", "Vxx", "vx", "Vxt", "Vtt", "vt"):
                a, b = getattr(f, nm), getattr(o, nm)
                if a.size:
                    assert a.shape == b.shape
                    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (t, nm)
        assert np.abs(rpar.condensed_solution() - opar.condensed_solution()).max() <= tol * sc
        rpar.collapseFeedback()
        opar.collapseFeedback()
        K0, K0o = rpar.datas(0).fb, opar.datas(0).fb
        assert np.abs(K0 - K0o).max() <= tol * max(1.0, np.abs(K0o).max())


def test_block_tridiagonal_solves(oracle):
    rng = np.random.default_rng(0)
    dims = [3, 5, 5, 4, 6]
    diag = [(g @ g.T + np.eye(n)) * (1 if i % 2 else -1)
            for i, n in enumerate(dims) for g in [rng.standard_normal((n, n))]]
    sup = [rng.standard_normal((dims[i], dims[i + 1])) for i in range(len(dims) - 1)]
    sub = [s.T.copy() for s in sup]
    rhs = [rng.standard_normal(n) for n in dims]
    for down in (False, True):
        ok1, x1 = ref.block_tridiag_solve(sub, diag, sup, rhs, down)
        ok2, x2 = oracle.block_tridiag_solve(sub, diag, sup, rhs, down)
        assert ok1 and ok2
        assert max(np.abs(a - b).max() for a, b in zip(x1, x2)) <= 1e-13


def test_stage_dense_solver_oracle_equals_reference_code():
    """oracle/dense_riccati.py (numpy restatement of gar/dense-kernel.hpp, dense-riccati.hxx) against the reference's
    RiccatiSolverDense itself: solution, [K; Z; L; Y] rows of ff / fb / ft, Pxx / px / Pxt / Ptt / pt of every stage,
    kkt0, thGrad / thHess.  (Terminal knot with nu = nc = 0, as in the reference's test and benchmark: with a
    constrained terminal knot the reference factorises a matrix with zero rows, see oracle/dense_riccati.py.)"""
    from oracle.dense_riccati import RiccatiSolverDense as OracleDense
    rng = np.random.default_rng(21)
    par, theta = _parametric()
    cases = [(synth.generate_lq_problem(5, np.zeros(36), 12, 36, 12, mode="W"), 1e-12, None, 1e-11),
             (synth.generate_lq_problem(6, np.zeros(8), 20, 8, 3, mode="F"), 1e-12, None, 1e-9),
             (par, 1e-12, theta, 1e-10)]
    c = _constrained(6, 3, 4, 9, 31)
    last = c.stages[-1]
    from aligator_amd.lqr import LqrKnot, LqrProblem
    term = LqrKnot(last.nx, 0, 0)
    term.Q[...], term.q[...] = last.Q, last.q
    cp = LqrProblem(c.stages[:-1] + [term], c.nc0)
    cp.G0[...], cp.g0[...] = c.G0, c.g0
    cases.append((cp, 1e-7, None, 1e-10))
    for prob, mu, th, tol in cases:
        rs = ref.RiccatiSolverDense(ref.Problem(prob))
        assert rs.backward(mu)
        rsol = rs.forward(th)
        o = OracleDense(prob, terminal_leading_block=False)
        o.backward(mu)
        osol = lqrInitializeSolution(prob)
        o.forward(*osol, th)
        sc = pc.scale_of(osol)
        for a, b in zip(rsol, osol):
            assert pc.maxdiff(a, b) <= tol * sc
        for t in range(prob.horizon + 1):
            f, d = rs.datas(t), o.stage_factors[t]
            if t < prob.horizon:
                for a, b in ((f.ff, d.ff), (f.fb, d.fb), (f.ft, d.ft)):
                    if a.size:
                        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), t
            for a, b in ((f.Pxx, o.Pxx[t]), (f.px, o.px[t]), (f.Pxt, o.Pxt[t]), (f.Ptt, o.Ptt[t]), (f.pt, o.pt[t])):
                if a.size:
                    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), t
        for a, b in zip(rs.initial(), (o.kkt0_ff, o.kkt0_fth, o.thGrad, o.thHess)):
            if a.size:
                assert np.abs(a - np.asarray(b).reshape(a.shape)).max() <= tol * max(1.0, np.abs(b).max())

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


# ---- rows a10 / f4 (cycleAppend) and a17 (lqrComputeKktError) against the reference's own code ----------------------
def _rotate(prob, new):
    """The caller's side of an MPC cycle (solvers/proxddp/workspace.hxx:122-126)."""
    N = prob.horizon
    prob.stages[:N] = prob.stages[1:N] + [new]


@pytest.mark.parametrize("nx,nu,nc", [(8, 4, 0), (6, 3, 2)])
def test_mpc_cycle_reference_code_vs_oracle_and_kernels(oracle, nx, nu, nc):
    """ProximalRiccatiSolver::cycleAppend (proximal-riccati.hxx:79-86) driven the way WorkspaceTpl::cycleAppend and
    solver-proxddp.hxx:208 drive it, on the reference's compiled code: after each cycle (more cycles than stages)
      reference(cycled) == reference(fresh solver on the rotated problem)   -- what the contract of the call is,
      reference(cycled) == oracle on the rotated problem                    -- the oracle's pin,
      reference(cycled) == the kernel sources on the emulator through the C ABI's ring (gar_hip_cycle_append),
    solution and every stage's gains and value function; and right after cycleAppend, before the next backward, the
    rotated factors: stage t holds what stage t+1 held, the last-but-one factor is re-created (zero gains), thGrad / thHess
    are zero."""
    import os
    import subprocess
    from aligator_amd.gar import ProximalRiccatiSolver
    here = os.path.dirname(os.path.abspath(__file__))
    emu = os.path.join(here, "emu", "_build", "libgar_hip_emu.so")
    subprocess.run(["make", "-s", "-C", os.path.join(here, "emu")], check=True)
    rng = np.random.default_rng(5 + nx)
    horz, mu, tol = 5, 1e-8, 1e-10
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    rp = ref.Problem(prob)
    rs = ref.ProximalRiccatiSolver(rp)
    ks = ProximalRiccatiSolver(prob, lib_path=emu)
    assert rs.backward(mu) and ks.backward(mu)
    rs.forward()
    for c in range(horz + 2):
        before = [rs.datas(t) for t in range(horz + 1)]
        new = synth.generate_knot(rng, nx, nu, nc=nc, mode="W")
        _rotate(prob, new)
        rp.cycle(new)
        rs.cycleAppend()
        ks.cycleAppend(new)
        for t in range(horz - 1):                                  # rotate_vec_left(datas, 0, 1)
            for nm in ("ff", "fb", "Vxx", "vx"):
                assert np.array_equal(getattr(rs.datas(t), nm), getattr(before[t + 1], nm)), (c, t, nm)
        for nm in ("ff", "fb"):                                    # StageFactor re-created: its constructor zeroes the
            assert not getattr(rs.datas(horz - 1), nm).any(), (c, nm)   # gains (riccati-kernel.hxx:46-49), NOT Vxx / vx
        for nm in ("ff", "fb", "Vxx", "vx"):                       # (CostToGo's leaves them uninitialised, .hpp:43-52)
            assert np.array_equal(getattr(rs.datas(horz), nm), getattr(before[horz], nm))
        assert not any(a.any() for a in rs.initial()[1:])          # thGrad, thHess (and kkt0.mat) zeroed
        assert rs.backward(mu) and ks.backward(mu)
        rsol = rs.forward()
        ksol = lqrInitializeSolution(prob)
        assert ks.forward(*ksol)
        fresh = ref.ProximalRiccatiSolver(ref.Problem(prob))
        assert fresh.backward(mu)
        fsol = fresh.forward()
        _, osol, oref = pc.oracle_serial(prob, mu)
        sc = pc.scale_of(oref)
        for a, f, o, k in zip(rsol, fsol, oref, ksol):
            assert pc.maxdiff(a, f) == 0.0, c                      # same code, same data: bitwise
            assert pc.maxdiff(a, o) <= tol * sc, c
            assert pc.maxdiff(a, k) <= 1e-9 * sc, c
        for t in range(horz + 1):
            r, o, k = rs.datas(t), osol.datas(t), ks.datas[t]
            for nm in ("ff", "fb", "Vxx", "vx"):
                a = getattr(r, nm)
                if not a.size:
                    continue
                b = getattr(o, nm)
                kk = getattr(k, nm) if nm in ("ff", "fb") else getattr(k.vm, nm)
                assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (c, t, nm)
                assert np.abs(a - np.asarray(kk).reshape(a.shape)).max() <= 1e-9 * max(1.0, np.abs(a).max()), (c, t, nm)


@pytest.mark.parametrize("nx,nu,nc,kernel", [(8, 4, 0, "wave_leg<8,4>"), (6, 3, 0, "wave_leg<8,4>"), (8, 4, 3, None)])
def test_parallel_cycle_reference_code_vs_oracle_and_kernels(oracle, nx, nu, nc, kernel):
    """ParallelRiccatiSolver::cycleAppend (parallel-solver.hxx:246-258: drop every parameterisation, initialise
    again) on the reference's compiled code: the cycled solver equals the oracle's leg-parallel solve of the rotated
    problem, the product's host mirror on the emulator (which re-parameterises the caller's problem the same way),
    and the RAW C ABI's gar_hip_cycle_append in leg mode ("just reinitialise everything": buffers rebuilt, the whole
    problem uploaded again) -- a shape with its own leg kernels, one padded inside the library, one with constraints
    folded onto them."""
    import os
    import subprocess
    from aligator_amd.gar import BatchedRiccatiSolver, ParallelRiccatiSolver
    here = os.path.dirname(os.path.abspath(__file__))
    emu = os.path.join(here, "emu", "_build", "libgar_hip_emu.so")
    subprocess.run(["make", "-s", "-C", os.path.join(here, "emu")], check=True)
    rng = np.random.default_rng(23 + nx + nc)
    horz, legs, mu = 11, 3, 1e-7 if nc else 1e-9
    tol = 1e-8 if nc else 1e-10
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horz, nx, nu, nc=nc, mode="W")
    plain = prob.copy()                                            # never parameterised: what the raw ABI is given
    rp = ref.Problem(prob)
    rpar = ref.ParallelRiccatiSolver(rp, legs)
    kprob = prob.copy()
    kpar = ParallelRiccatiSolver(kprob, legs, lib_path=emu)
    if kernel:
        assert kpar._impl.kernel_name == kernel
    raw = BatchedRiccatiSolver([k.dims for k in plain.stages], plain.nc0, 1, legs, lib_path=emu)
    raw.upload([plain])
    assert rpar.backward(mu) and kpar.backward(mu) and raw.backward(mu)
    rpar.forward()
    for c in range(3):
        new = synth.generate_knot(rng, nx, nu, nc=nc, mode="W")
        for p in (prob, kprob, plain):
            _rotate(p, new.copy())
        rp.cycle(new)
        rpar.cycleAppend()
        kpar.cycleAppend(new)
        raw.cycle_append(new.dims)                                 # the C entry point itself, leg mode
        raw.upload([plain])
        assert rpar.backward(mu) and kpar.backward(mu) and raw.backward(mu)
        rsol = rpar.forward()
        ksol = lqrInitializeSolution(plain)
        assert kpar.forward(*ksol)
        assert raw.forward()
        opar = oracle.ParallelRiccatiSolver(pc.to_oracle(plain.copy()), legs)
        opar.backward(mu)
        osol = lqrInitializeSolution(plain)
        opar.forward(*osol)
        sc = pc.scale_of(osol)
        for a, b, k, w in zip(rsol, osol, ksol, raw.solution(0)):
            assert pc.maxdiff(a, b) <= tol * sc, c
            assert pc.maxdiff(a, k) <= 10 * tol * sc, c
            assert pc.maxdiff(a, w) <= 10 * tol * sc, c
        # the reference re-parameterised the rotated problem: the same knots carry nth = nx in the mirror's problem
        for t in range(horz + 1):
            assert kprob.stages[t].nth == rpar.nth(t), (c, t)


def test_kkt_error_equals_reference_code(oracle):
    """lqrComputeKktError (gar/utils.hxx:88-182) compiled from the reference against the two host mirrors' restatement
    (aligator_amd/lqr.py, numpy) and the oracle's (oracle/gar_oracle.c), on trajectories that are NOT solutions (every
    term of every residual is non-zero), with and without theta, constrained and not, nu = 0 at the terminal knot."""
    from aligator_amd.lqr import lqrComputeKktError
    rng = np.random.default_rng(31)
    par, theta = _parametric()
    for prob, th, mu in ((synth.generate_lq_problem(rng, rng.standard_normal(8), 9, 8, 4, mode="F"), None, 1e-7),
                         (_constrained(6, 3, 4, 7, 11), None, 1e-3),
                         (par, theta, 1e-5),
                         (synth.short_horizon_problem(8), None, 0.0)):
        sol = lqrInitializeSolution(prob)
        for group in sol:
            for a in group:
                a[...] = rng.standard_normal(a.shape)
        r = ref.kkt_error(ref.Problem(prob), *sol, mu, th)
        mine = lqrComputeKktError(prob, *sol, mueq=mu, theta=th)
        theirs = oracle.lqr_kkt_error(pc.to_oracle(prob), *sol, mu, th)
        assert min(r) > 1e-3 or prob.stages[0].nc == 0            # the comparison is not 0 == 0
        for a, b, c in zip(r, mine, theirs):
            assert abs(a - b) <= 1e-13 * max(1.0, abs(a)), (r, mine)
            assert abs(a - c) <= 1e-13 * max(1.0, abs(a)), (r, theirs)


def test_dense_cycle_reference_code_vs_oracle_and_kernels():
    """RiccatiSolverDense::cycleAppend (dense-riccati.hxx:118-146) on the reference's compiled code, cycled more
    often than there are stages: equal to the numpy restatement solving the rotated problem afresh and to the dense
    kernels on the emulator through the C ABI's ring."""
    import os
    import subprocess
    from oracle.dense_riccati import RiccatiSolverDense as OracleDense
    from aligator_amd.gar import RiccatiSolverDense
    here = os.path.dirname(os.path.abspath(__file__))
    emu = os.path.join(here, "emu", "_build", "libgar_hip_emu.so")
    subprocess.run(["make", "-s", "-C", os.path.join(here, "emu")], check=True)
    rng = np.random.default_rng(41)
    nx, nu, horz, mu = 6, 3, 4, 1e-9
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    rp = ref.Problem(prob)
    rs = ref.RiccatiSolverDense(rp)
    ks = RiccatiSolverDense(prob, lib_path=emu)
    assert rs.backward(mu) and ks.backward(mu)
    rs.forward()
    for c in range(horz + 2):
        new = synth.generate_knot(rng, nx, nu, mode="W")
        _rotate(prob, new)
        rp.cycle(new)
        rs.cycleAppend()
        ks.cycleAppend(new)
        assert rs.backward(mu) and ks.backward(mu)
        rsol = rs.forward()
        ksol = lqrInitializeSolution(prob)
        assert ks.forward(*ksol)
        o = OracleDense(prob, terminal_leading_block=False)
        o.backward(mu)
        osol = lqrInitializeSolution(prob)
        o.forward(*osol)
        sc = pc.scale_of(osol)
        for a, b, k in zip(rsol, osol, ksol):
            assert pc.maxdiff(a, b) <= 1e-10 * sc, c
            assert pc.maxdiff(a, k) <= 1e-9 * sc, c
        for t in range(horz + 1):
            f = rs.datas(t)
            for a, b in ((f.Pxx, o.Pxx[t]), (f.px, o.px[t])):
                assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), (c, t)


def test_round6_soak_draw_arbitrated_by_the_reference_code():
    """The one failing draw of the round-6 soak (SOAK_SEED 6103, focus constrained, inner seed 167330036: (6, 3, 2),
    N = 68, serial, mu = 1.12e-9 -- 12 % above the mu the soak stops comparing constrained stage factors at -- on the
    any-dimension kernels): `ff` of stage 9 (D != 0, R scaled by 1e-3: Bunch-Kaufman interchanges, the reduced KKT matrix
    is conditioned like 4.7e11) 1.0e-6 from the oracle's, of its scale, against the soak's bar of 1e-6.  Three-way, with
    the reference's own compiled code: kernels - reference 1.6e-6, oracle - reference 5.9e-7 on that block, 1e-11 or
    less on every other stage: what the conditioning of that one matrix allows any two factorisations of it.  Held here:
    no block of the kernels' factors is farther from the reference's than 10 x the oracle's is (floor 1e-9)."""
    import os
    from soak_draws import draws
    from aligator_amd.gar import ProximalRiccatiSolver
    for i, d in enumerate(draws("6103", "constrained", version=2)):
        if d["seed"] == 167330036:
            break
        assert i < 1000
    prob, mu = d["prob"], d["mu"]
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    emu = os.path.join(here, "emu", "_build", "libgar_hip_emu.so")
    subprocess.run(["make", "-s", "-C", os.path.join(here, "emu")], check=True)
    s = ProximalRiccatiSolver(prob, lib_path=emu)
    assert s.kernel_name == "generic" and s.backward(mu)
    _, osol, _ = pc.oracle_serial(prob, mu)
    rs = ref.ProximalRiccatiSolver(ref.Problem(prob))
    assert rs.backward(mu)
    worst = 0.0
    for t in range(prob.horizon + 1):
        h, o, r = s.datas[t], osol.datas(t), rs.datas(t)
        for nm in ("ff", "fb"):
            a, b, c = getattr(h, nm), getattr(o, nm), getattr(r, nm)
            if a.size:
                sc = max(1.0, np.abs(c).max())
                hr, orr = np.abs(a - c).max() / sc, np.abs(b - c).max() / sc
                worst = max(worst, hr)
                assert hr <= 10.0 * max(orr, 1e-9), (t, nm, hr, orr)
    assert 1e-7 < worst < 1e-5        # (the draw is the ill-conditioned one it was logged as)

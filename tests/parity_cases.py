"""Parity checks shared by the emulator tests (CPU, small sizes) and the GPU
tests (real library, reference sizes).  Each check mirrors a test of the
reference (tests/gar/riccati.cpp, tests/gar/parallel.cpp) and compares the HIP
path (through the C ABI) with the CPU oracle on the same seeded input.

Stated fp64 tolerances: relative 1e-9 on the
well-conditioned generator "W", 1e-6 on the reference-faithful generator "F"
(the reference's own bar at nx=36, tests/gar/riccati.cpp:138), relative to the
largest multiplier / value-function entry of the oracle solution.
"""
import os

import numpy as np

from aligator_amd import synth
from aligator_amd.gar import (BatchedRiccatiSolver, ParallelRiccatiSolver,
                              ProximalRiccatiSolver, RiccatiSolverDense, lqrComputeKktError,
                              lqrInitializeSolution)
from oracle import oracle as ora
from oracle.dense_riccati import RiccatiSolverDense as OracleDense

TOL = {"W": 1e-9, "F": 1e-6}


def to_oracle(prob):
    return ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)


def maxdiff(A, B):
    return max([0.0] + [float(np.max(np.abs(a - b))) for a, b in zip(A, B) if a.size])


def scale_of(sol):
    return max(1.0, max(float(np.abs(v).max()) for part in sol for v in part if v.size))


def oracle_serial(prob, mueq, theta=None):
    op = to_oracle(prob)
    s = ora.ProximalRiccatiSolver(op)
    assert s.backward(mueq)
    sol = lqrInitializeSolution(prob)
    s.forward(*sol, theta)
    return op, s, sol


def compare_factors(hip_datas, ora_solver, N, tol, names=("ff", "fb", "fth"),
                    vnames=("Vxx", "vx", "Vxt", "Vtt", "vt")):
    """Gains and value function of every stage, relative to each block's scale."""
    for t in range(N + 1):
        f, o = hip_datas[t], ora_solver.datas(t)
        for nm in names:
            a, b = getattr(f, nm), getattr(o, nm)
            if a.size:
                assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (t, nm)
        for nm in vnames:
            a, b = getattr(f.vm, nm), getattr(o, nm)
            if a.size:
                assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (t, nm)
        # StageFactor::kktMat (formed on the device on request: gar_hip_get_kkt).  (nu = 0: the reference's
        # terminalSolve never assembles it, :146-149)
        km = getattr(f, "kktMat", None)
        if km is not None and km.size and f.nu > 0 and "fb" in names:
            b = o.kktMat
            assert np.abs(km - b).max() <= max(tol, 1e-9) * max(1.0, np.abs(b).max()), (t, "kktMat")


CONDITIONING_MARGIN = 10.0   # x what three independent CPU solves of the same problem disagree by


def check_serial(prob, mueq, tol, lib_path=None, theta=None, kkt_tol=None, factors=True, conditioned=False):
    """conditioned (constrained problems with a small mu; theta = None): each part of the solution is held to
    max(tol, CONDITIONING_MARGIN x the disagreement of the oracle and LAPACK on the dense KKT matrix) -- see
    conditioning_bound."""
    solver = ProximalRiccatiSolver(prob, lib_path=lib_path)
    assert solver.backward(mueq)
    sol = lqrInitializeSolution(prob)
    assert solver.forward(*sol, theta)
    _, osol, ref = oracle_serial(prob, mueq, theta)
    sc = scale_of(ref)
    tols = [tol] * 4
    if conditioned and theta is None:
        bound, _ = conditioning_bound(prob, mueq, ref)
        tols = [max(tol, CONDITIONING_MARGIN * b) for b in bound]
    for A, B, tl in zip(sol, ref, tols):
        assert maxdiff(A, B) <= tl * sc
    if kkt_tol is not None:
        assert max(lqrComputeKktError(prob, *sol, mueq=mueq, theta=theta)) <= kkt_tol
    if factors:
        compare_factors(solver.datas, osol, prob.horizon, tol)
        ff, fth, g, H = solver._impl.initial(0)
        for a, b in ((ff, osol.kkt0_ff), (fth, osol.kkt0_fth), (g, osol.thGrad), (H, osol.thHess)):
            if a.size:
                assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    return solver, sol, ref


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if b.size else 0.0


def check_dense(prob, mueq, tol, lib_path=None, theta=None, kkt_tol=None):
    """RiccatiSolverDense (tests/gar/riccati.cpp:141-155): the HIP stage-dense solver against its
    oracle (oracle/dense_riccati.py) -- trajectory, every stage's ff / fb / ft rows [K; Z; L; Y],
    Pxx, px, Pxt, Ptt, pt, kkt0, thGrad, thHess -- and against the Riccati oracle's trajectory."""
    solver = RiccatiSolverDense(prob, lib_path=lib_path)
    assert solver.kernel_name == "dense"
    assert solver.backward(mueq)
    sol = lqrInitializeSolution(prob)
    assert solver.forward(*sol, theta)
    o = OracleDense(prob)
    o.backward(mueq)
    ref = lqrInitializeSolution(prob)
    o.forward(*ref, theta)
    sc = scale_of(ref)
    for A, B in zip(sol, ref):
        assert maxdiff(A, B) <= tol * sc
    _, _, ric = oracle_serial(prob, mueq, theta)
    for A, B in zip(sol, ric):
        assert maxdiff(A, B) <= 10 * tol * sc
    if kkt_tol is not None:
        assert max(lqrComputeKktError(prob, *sol, mueq=mueq, theta=theta)) <= kkt_tol
    for t in range(prob.horizon + 1):
        f, d = solver.datas[t], o.stage_factors[t]
        assert f.ff.shape == d.ff.shape and f.fb.shape == d.fb.shape and f.fth.shape == d.ft.shape
        for a, b in ((f.ff, d.ff), (f.fb, d.fb), (f.fth, d.ft), (f.vm.Vxx, o.Pxx[t]), (f.vm.vx, o.px[t]),
                     (f.vm.Vxt, o.Pxt[t]), (f.vm.Vtt, o.Ptt[t]), (f.vm.vt, o.pt[t])):
            assert _rel(a, b) <= tol, t
        assert np.array_equal(solver.getFeedback(t), f.fb) and np.array_equal(solver.getFeedforward(t), f.ff)
    for a, b in ((solver.kkt0.ff, o.kkt0_ff), (solver.kkt0.fth, o.kkt0_fth), (solver.thGrad, o.thGrad),
                 (solver.thHess, o.thHess)):
        assert _rel(a, b) <= tol
    return solver, sol, ref


def conditioning_bound(prob, mueq, ref=None):
    """What cond * eps allows on THIS problem, per part (x, u, v, lambda), relative to the solution's scale: the
    disagreement of two independent CPU solves -- the Riccati oracle (serial) and LAPACK on the global dense
    KKT matrix laid out like the reference's test helper (oracle/dense_kkt.py).  Constrained problems with a
    small mu carry multipliers of order 1/mu whose last digits no two factorisations share (the two soak
    draws replayed by test_soak_failures_replayed_*: x, u agree to 1e-15, v and lambda to 1e-7 ... 2e-6
    between ANY two of oracle-serial / oracle-leg / LAPACK / HIP)."""
    from oracle import dense_kkt
    if ref is None:
        _, _, ref = oracle_serial(prob, mueq)
    lap = dense_kkt.dense_solve(prob, mueq)
    sc = scale_of(ref)
    return [maxdiff(a, b) / sc for a, b in zip(ref, lap)], lap


def check_parallel(prob, mueq, nthreads, tol, lib_path=None, max_refine=10, rounds=0, rng=None, conditioned=False,
                   report=None, devices=None):
    """tests/gar/parallel.cpp:185-245 (parallel_solver_class).
    conditioned: the tolerance of each part of the solution is max(tol, CONDITIONING_MARGIN x what the problem's
    conditioning allows), measured on the problem itself (conditioning_bound, and the oracle's own leg-parallel vs serial
    solutions) instead of a floor on mu in the caller.  report: a dict that receives every pairwise figure."""
    _, _, ref = oracle_serial(prob, mueq)
    pprob = prob.copy()
    # devices: the legs split over several devices inside the one solver object (gar_hip_multi_create)
    par = ParallelRiccatiSolver(pprob, nthreads, lib_path=lib_path, devices=devices)
    par.maxRefinementSteps = max_refine
    sol = lqrInitializeSolution(pprob)
    assert par.backward(mueq)
    assert par.forward(*sol)
    sc = scale_of(ref)
    # per-stage factors against the oracle's own leg-parallel solver
    op = to_oracle(prob)
    opar = ora.ParallelRiccatiSolver(op, nthreads)
    opar.maxRefinementSteps = max_refine
    opar.backward(mueq)
    tols, ktol, ftol = [tol] * 4, tol, max(tol, 1e-8)
    if conditioned or report is not None:
        osol = lqrInitializeSolution(prob)
        opar.forward(*osol)
        bound, lap = conditioning_bound(prob, mueq, ref)
        leg_vs_serial = [maxdiff(a, b) / sc for a, b in zip(osol, ref)]
        okkt = max(lqrComputeKktError(prob, *osol, mueq=mueq)) / sc
        if conditioned:
            m = CONDITIONING_MARGIN
            tols = [max(tol, m * b, m * l) for b, l in zip(bound, leg_vs_serial)]
            ktol = max(tol, m * okkt)
            ftol = max(ftol, m * max(bound), m * max(leg_vs_serial))
        if report is not None:
            pairs = {"hip_leg-oracle_leg": (sol, osol), "hip_leg-oracle_serial": (sol, ref), "hip_leg-lapack": (sol, lap),
                     "oracle_leg-oracle_serial": (osol, ref), "oracle_leg-lapack": (osol, lap),
                     "oracle_serial-lapack": (ref, lap)}
            report.update({k: [maxdiff(a, b) / sc for a, b in zip(*v)] for k, v in pairs.items()})
            report.update(scale=sc, tolerances=tols, kkt_tolerance=ktol, kkt_oracle_leg=okkt,
                          kkt_hip_leg=max(lqrComputeKktError(pprob, *sol, mueq=mueq)) / sc)
    assert max(lqrComputeKktError(pprob, *sol, mueq=mueq)) <= ktol * sc   # parallel.cpp:221
    assert maxdiff(sol[0], ref[0]) <= tols[0] * sc                         # :234
    assert maxdiff(sol[3], ref[3]) <= tols[3] * sc                         # :235
    if conditioned:
        assert maxdiff(sol[1], ref[1]) <= tols[1] * sc and maxdiff(sol[2], ref[2]) <= tols[2] * sc
    # the knots were re-parameterised in place like the reference does
    b0, e0 = 0, (prob.horizon + 1) // nthreads
    assert pprob.stages[b0].nth == prob.stages[e0 - 1].nx2
    assert np.array_equal(pprob.stages[e0 - 1].Gx, pprob.stages[e0 - 1].A.T)
    # (stage factors of constrained problems below mu ~ 1e-9: two Bunch-Kaufman runs on the same reduced KKT matrix,
    # conditioned like 1/mu, differ block by block by more than the solution does -- the oracle and LAPACK do:
    # the solution-level checks above stand alone there, as in check_serial's `factors` switch of the soak)
    if not (conditioned and mueq <= 1e-9 and any(k.nc > 0 for k in prob.stages)):
        compare_factors(par.datas, opar, prob.horizon, ftol)
        par.collapseFeedback()
        opar.collapseFeedback()
        K0, K0o = par.getFeedback(0), opar.datas(0).fb
        assert np.abs(K0 - K0o).max() <= ftol * max(1.0, np.abs(K0o).max())
    for _ in range(rounds):                                              # :238-244
        synth.randomly_modify_problem(rng, pprob)
        par.backward(mueq)
        par.forward(*sol)
        assert max(lqrComputeKktError(pprob, *sol, mueq=mueq)) <= ktol * sc
    return par


def check_batched(probs, mueq, tol, lib_path=None, num_legs=1):
    """The batch axis: each problem of a batch equals its own serial oracle solve."""
    p0 = probs[0]
    dims = [k.dims for k in p0.stages]
    s = BatchedRiccatiSolver(dims, p0.nc0, batch=len(probs), num_legs=num_legs, lib_path=lib_path)
    s.upload(probs)
    assert s.backward(mueq)
    assert s.forward()
    for b, prob in enumerate(probs):
        _, _, ref = oracle_serial(prob, mueq)
        sol = s.solution(b)
        sc = scale_of(ref)
        for A, B in zip(sol, ref):
            assert maxdiff(A, B) <= tol * sc, b
    return s


def _leg_solution(probs, legs, mueq, lib_path=None, refine=None, threshold=1e-10):
    dims = [k.dims for k in probs[0].stages]
    s = BatchedRiccatiSolver(dims, probs[0].nc0, batch=len(probs), num_legs=legs, lib_path=lib_path)
    if refine is not None:
        s.set_refinement(threshold, refine)
    s.upload(probs)
    assert s.backward(mueq) and s.forward()
    return s, [s.solution(b) for b in range(len(probs))]


def check_condensed_block_inverse_fallback(lib_path=None):
    """A leg-start value function whose unpivoted LDL^T fails the first Bunch-Kaufman test
    (|a_kk| < alpha * max|a_ik|).  Cyclic reduction only ever inverts definite blocks and does not
    pivot; the elimination-chain kernel follows the reference and hands that block to the generic
    device Bunch-Kaufman (interchanges / 2x2 pivots).  Both must agree with the oracle."""
    nx, nu, N, legs = 8, 4, 11, 3
    prob = synth.generate_lq_problem(77, np.ones(nx), N, nx, nu, mode="W")
    t0 = 8                                     # first stage of the final leg
    M = np.diag(np.r_[1.0, 100.0, np.full(nx - 2, 10.0)])
    M[0, 1] = M[1, 0] = 5.0                    # SPD, |a_00| = 1 < 0.64 * 5
    prob.stages[t0].Q[...] = 1e4 * M
    osol = ora.ProximalRiccatiSolver(to_oracle(prob))
    osol.backward(1e-10)
    V = osol.datas(t0).Vxx
    alpha = (1 + np.sqrt(17)) / 8
    assert abs(V[0, 0]) < alpha * np.abs(V[1:, 0]).max()      # the premise of this test
    for refine in (0, 3):
        s, sol = _leg_solution([prob], legs, 1e-10, lib_path, refine=refine)
        assert s.kernel_name.startswith("wave_leg<") and s.num_failed() == 0
        _, _, ref = oracle_serial(prob, 1e-10)
        for A, B in zip(sol[0], ref):
            assert maxdiff(A, B) <= 1e-9 * scale_of(ref)
    # and through the elimination-chain kernel, whose last block is the same Vxx
    import os
    os.environ["GAR_HIP_CONDENSED"] = "chain"
    try:
        s, sol = _leg_solution([prob], legs, 1e-10, lib_path)
    finally:
        del os.environ["GAR_HIP_CONDENSED"]
    for A, B in zip(sol[0], ref):
        assert maxdiff(A, B) <= 1e-9 * scale_of(ref)


def check_leg_kernels_bunch_kaufman_fallback(lib_path=None, nthreads=3, horz=11):
    """Stages whose Rhat makes Bunch-Kaufman interchange (bunchkaufman.hpp:61-83), in leg mode:
    the plain part takes the generic device Bunch-Kaufman and the parameter part (the second
    right-hand-side set; wave B of the two-wave kernel) solves Kth with those pivoted factors.
    A failing factorisation must be reported (riccati-kernel.hxx:239-241)."""
    import pytest
    nx, nu = 8, 4
    prob = synth.generate_lq_problem(11, np.zeros(nx), horz, nx, nu, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = np.array([[1e-3, 2.0, 0.1, 0.0], [2.0, 1e-3, 0.0, 0.1],
                             [0.1, 0.0, 3.0, 0.2], [0.0, 0.1, 0.2, 4.0]])
        k.B[...] *= 1e-2
    op = to_oracle(prob)
    opar = ora.ParallelRiccatiSolver(op, nthreads)
    opar.backward(1e-12)
    pivots = [ora.BunchKaufman(opar.datas(t).Rhat).pivots for t in range(horz)]
    assert any(not np.array_equal(p, np.arange(nu)) for p in pivots), "test must force a pivot"
    par = check_parallel(prob, 1e-12, nthreads, 1e-9, lib_path)
    assert par._impl.kernel_name == "wave_leg<8,4>"
    bad = synth.generate_lq_problem(3, np.zeros(nx), horz, nx, nu, mode="W")
    for k in bad.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    with pytest.raises(RuntimeError, match="LDL"):
        ParallelRiccatiSolver(bad, nthreads, lib_path=lib_path).backward(1e-10)


def check_constrained_pivoting(lib_path=None, shapes=((8, 4, 4, 6, 1e-6), (16, 8, 8, 5, 1e-7))):
    """Constrained stages whose reduced KKT matrix makes Bunch-Kaufman pivot: (a) small Rhat against
    D entries of order one -> every pivot a 2x2 block; (b) mixed -> 1x1 interchanges.  (The
    reference's generator, tests/gar/test_util.cpp:42-43, leaves D = 0: its KKT matrix is block
    diagonal and never pivots.)  Factors, kkt0 and the solution against the oracle."""
    for variant in ("all2x2", "mixed"):
        for (nx, nu, nc, horz, mu) in shapes:
            rng = np.random.default_rng(11)
            prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
            for k in prob.stages[:-1]:
                k.D[...] = rng.uniform(-1, 1, k.D.shape)
                if variant == "all2x2":
                    k.R[...] *= 1e-3
                    k.S[...] *= 1e-3
                    k.B[...] = 0.0
                else:
                    k.R[...] *= 1e-2
            _, osol, _ = oracle_serial(prob, mu)
            n2 = nsw = 0
            for t in range(horz):
                K = np.block([[osol.datas(t).Rhat, prob.stages[t].D.T],
                              [prob.stages[t].D, -mu * np.eye(nc)]])
                piv = ora.BunchKaufman(K).pivots
                n2 += int((piv < 0).sum())
                nsw += int(((piv >= 0) & (piv != np.arange(piv.size))).sum())
            if variant == "all2x2":
                assert n2 > 0, "test must force 2x2 pivots"
            elif nx <= 16:
                assert nsw > 0, "test must force interchanges"
            solver, _, _ = check_serial(prob, mu, 1e-8, lib_path)
            assert solver.kernel_name == f"wave<{nx},{nu},{nc}>"



def check_constrained_decoupled(lib_path=None, shapes=((8, 4, 4, 7, 1e-6), (16, 8, 8, 5, 1e-8), (36, 12, 32, 4, 1e-9)),
                                tol=1e-8):
    """The decoupled constrained stage (D = 0: gar_wave2.hpp, NC > 0) with a DENSE random C -- the
    reference's generator has C = [I 0] (tests/gar/test_util.cpp:42-43), which would hide an index slip in
    Vxx += C^T Z -- and problems that alternate between knots with D = 0 and knots with D != 0, so that the
    sweep switches between the two stage implementations (deferred Vxx flush on one side, own flush on the
    other).  Factors, kkt0 and the solution against the oracle."""
    decoupled_stages = coupled_stages = 0
    for variant in ("all_decoupled", "alternating"):
        for (nx, nu, nc, horz, mu) in shapes:
            rng = np.random.default_rng(77 + nx)
            prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
            for t, k in enumerate(prob.stages):
                k.C[...] = rng.uniform(-1, 1, k.C.shape)
                if variant == "alternating" and t % 2 == 1 and t < horz:
                    k.D[...] = rng.uniform(-1, 1, k.D.shape)
            solver, _, _ = check_serial(prob, mu, tol, lib_path)
            assert solver.kernel_name == f"wave<{nx},{nu},{nc}>"
            # kernel chain: decoupled stages down to the first knot that is not one (D != 0, or an Rhat on
            # which Bunch-Kaufman pivots), coupled stages from there to the first pivoting KKT matrix, the LDS
            # Bunch-Kaufman for the rest
            coupled, bk = solver._impl.constrained_bk_stages()
            assert coupled + bk <= horz, (nx, coupled, bk)
            if variant == "alternating":
                t_first = max(t for t in range(horz) if t % 2 == 1)
                assert coupled + bk >= t_first + 1, (nx, coupled, bk, t_first)
                coupled_stages += coupled
            decoupled_stages += horz - coupled - bk
    assert decoupled_stages > 0, "no stage ran decoupled: the test does not reach gar_wave2.hpp's NC > 0 path"
    assert coupled_stages > 0, "no stage ran coupled: the test does not reach gar_wave2.hpp's COUPLED path"


def check_constrained_legs_segments(lib_path=None, shapes=((8, 4, 4, 13, 3, 1e-6), (16, 8, 8, 11, 2, 1e-7)), tol=1e-8):
    """Leg mode on problems with COUPLED constraints (D != 0) -- the constrained segment legs of gar_cstr_seg.hpp: the
    serial constrained chain's three stage kernels over each leg's stage range + the parameter recursion with the
    reference's own Bunch-Kaufman of [Rhat D^T; D -mu I].  D on every knot; D on alternating knots (the chain switches
    between the decoupled and the coupled stage inside a leg); knots on which Bunch-Kaufman really pivots (the LDS
    Bunch-Kaufman stage inside a leg).  Solution, every stage's factors incl. fth / Vxt / Vtt / vt and the collapsed K0
    against the oracle's leg-parallel solver; a batch that mixes the fold's problems with these."""
    from aligator_amd.gar import BatchedRiccatiSolver
    reached_bk = reached_coupled = 0
    for variant in ("every_knot", "alternating", "pivoting"):
        for (nx, nu, nc, horz, legs, mu) in shapes:
            rng = np.random.default_rng(5 + nx)
            prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
            for t, k in enumerate(prob.stages[:-1]):
                k.C[...] = rng.uniform(-1, 1, k.C.shape)
                if variant != "alternating" or t % 2 == 1:
                    k.D[...] = rng.uniform(-1, 1, k.D.shape)
                if variant == "pivoting" and t % 3 == 0:
                    k.R[...] *= 1e-2
            par = check_parallel(prob, mu, legs, tol, lib_path, conditioned=True)
            assert par._impl.kernel_name == f"wave_leg<{nx},{nu}>+fold|wave_seg<{nx},{nu},{nc}>", par._impl.kernel_name
            coupled, bk = par._impl.constrained_bk_stages()
            reached_coupled += coupled
            reached_bk += bk
            assert coupled + bk > 0, "the problem was not swept by the constrained segment legs"
            if variant == "pivoting":
                _, osol, _ = oracle_serial(prob, mu)
                opar = ora.ParallelRiccatiSolver(to_oracle(prob), legs)
                opar.backward(mu)
                nsw = 0
                for t in range(horz):
                    K = np.block([[opar.datas(t).Rhat, prob.stages[t].D.T], [prob.stages[t].D, -mu * np.eye(nc)]])
                    piv = ora.BunchKaufman(K).pivots
                    nsw += int((piv != np.arange(piv.size)).sum())
                if nx <= 16:
                    assert nsw > 0 and bk > 0, ("test must reach the LDS Bunch-Kaufman stage inside a leg", nsw, bk)
    assert reached_coupled > 0 and reached_bk > 0
    # the leg ends through the stage chain itself (two rounds) instead of the leg-end kernel: the switch, per launch
    os.environ["GAR_HIP_CSTR_SEG_LEG_END"] = "0"
    try:
        nx, nu, nc, horz, legs, mu = shapes[0]
        rng = np.random.default_rng(17)
        prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
        for k in prob.stages[:-1]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
        par = check_parallel(prob, mu, legs, tol, lib_path, conditioned=True)
        assert sum(par._impl.constrained_bk_stages()) > 0
    finally:
        del os.environ["GAR_HIP_CSTR_SEG_LEG_END"]
    # one batch: a folded problem (D = 0), a coupled one, a folded one -- each family skips the other's problems
    nx, nu, nc, horz, legs, mu = shapes[0]
    rng = np.random.default_rng(3)
    p0 = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    p1 = p0.copy()
    for k in p1.stages[:-1]:
        k.D[...] = rng.uniform(-1, 1, k.D.shape)
    p2 = p0.copy()
    p2.stages[1].q[...] += 1.0
    probs = [p0, p1, p2]
    s = BatchedRiccatiSolver([k.dims for k in p0.stages], p0.nc0, batch=3, num_legs=legs, lib_path=lib_path)
    for rep in range(2):                                                  # (twice: the flags are reset per backward)
        s.upload(probs)
        assert s.backward(mu) and s.forward()
        for b, p in enumerate(probs):
            _, _, ref = oracle_serial(p, mu)
            sc = scale_of(ref)
            for A, B in zip(s.solution(b), ref):
                assert maxdiff(A, B) <= max(tol, 1e-7) * sc, (rep, b)
            opar = ora.ParallelRiccatiSolver(to_oracle(p), legs)
            opar.backward(mu)

            class D:
                def __getitem__(self, t, b=b):
                    return s.factor(t, b)
            compare_factors(D(), opar, horz, max(tol, 1e-7))
        probs = [p1, p0, p2]                                              # the coupled problem moves to another slot
    # mueq = 0 (or NaN) on knots whose solve divides by it (Z = C / mu at the terminal knot, riccati-kernel.hxx:146-149;
    # D = 0 knots: kktMat singular, the reference throws, :239-241): reported as a failed factorisation before anything
    # is launched -- the infinities used to reach the pivot searches downstream
    for bad_mu in (0.0, float("nan")):
        try:
            ok = s.backward(bad_mu)
        except RuntimeError as e:
            ok = False
            assert "Failed stage LDL factorization" in str(e)
        assert not ok
    assert s.backward(mu) and s.forward()


def check_second_bunch_kaufman_test(lib_path=None):
    """Stages whose Rhat fails the FIRST Bunch-Kaufman test (|a_kk| < alpha colmax) but passes the second
    (|a_kk| rowmax >= alpha colmax^2, bunchkaufman.hpp:63-75): Bunch-Kaufman keeps kp = k, the kernel
    must stay on its register LDL^T (checked out of line at that column only) -- and a variant where the
    second test fails too, which must take the generic device Bunch-Kaufman.  Counters: (stages that
    needed the second test, stages that really pivoted)."""
    import os
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, horz = 8, 4, 6
    keep = np.array([[1.0, 2.0, 0.0, 0.0], [2.0, 10.0, 3.0, 0.0], [0.0, 3.0, 20.0, 0.1], [0.0, 0.0, 0.1, 4.0]])
    pivot = keep.copy()
    pivot[2, 1] = pivot[1, 2] = 0.5                      # rowmax 2 < alpha * 4 / 1: interchange
    old = os.environ.get("GAR_HIP_BACKWARD")
    os.environ["GAR_HIP_BACKWARD"] = "wave"
    old_spd = os.environ.get("GAR_HIP_SPD_ACCEPT")
    try:
        # 2x2 pivots on the plain stage's pivoting path: at column 0 with its partner next to it, and at column 1
        # with the partner two rows down (interchange 2 <-> 3)
        two = np.array([[0.1, 5.0, 0.0, 0.0], [5.0, 0.2, 1.0, 0.0], [0.0, 1.0, 20.0, 0.1], [0.0, 0.0, 0.1, 4.0]])
        far = np.array([[10.0, 0.1, 0.1, 0.1], [0.1, 0.05, 0.2, 6.0], [0.1, 0.2, 8.0, 0.1], [0.1, 6.0, 0.1, 0.03]])
        assert (ora.BunchKaufman(two).pivots < 0).sum() == 2 and (ora.BunchKaufman(far).pivots < 0).sum() == 2
        assert ora.BunchKaufman(far).pivots[1] == -1 - 3          # partner row 3, moved next to the pivot
        # spd: a positive definite Rhat keeps the unpivoted LDL^T even where Bunch-Kaufman interchanges (the default,
        # wave_ldl_fast_neg_pre); "0": the reference's pivot rule literally.  `pivot` is positive definite, `two`
        # and `far` are indefinite: they take the device Bunch-Kaufman either way.
        assert np.linalg.eigvalsh(pivot).min() > 0 and np.linalg.eigvalsh(two).min() < 0 and np.linalg.eigvalsh(far).min() < 0
        for spd, Rm, want in (("0", keep, (horz, 0)), ("0", pivot, (horz, horz)), ("0", two, (horz, horz)), ("0", far, (horz, horz)),
                              ("1", keep, (horz, 0)), ("1", pivot, (horz, 0)), ("1", two, (horz, horz)), ("1", far, (horz, horz))):
            os.environ["GAR_HIP_SPD_ACCEPT"] = spd
            prob = synth.generate_lq_problem(21, np.zeros(nx), horz, nx, nu, mode="W")
            for k in prob.stages[:-1]:
                k.R[...] = Rm
                k.S[...] = 0.0
                k.B[...] = 0.0                           # Rhat = R at every stage
            piv = ora.BunchKaufman(Rm).pivots
            assert np.array_equal(piv, np.arange(nu)) == (want[1] == 0 and not (spd == "1" and Rm is pivot)), piv
            s = BatchedRiccatiSolver([k.dims for k in prob.stages], prob.nc0, batch=1, lib_path=lib_path)
            assert s.kernel_name == "wave<8,4>"
            s.upload([prob])
            assert s.backward(1e-12) and s.forward()
            assert s.slow_path_stages() == want
            _, osol, ref = oracle_serial(prob, 1e-12)
            for A, B in zip(s.solution(0), ref):
                assert maxdiff(A, B) <= 1e-9 * scale_of(ref)

            class D:
                def __getitem__(self, t):
                    return s.factor(t, 0)
            compare_factors(D(), osol, horz, 1e-9, names=("ff", "fb"), vnames=("Vxx", "vx"))
    finally:
        if old is None:
            del os.environ["GAR_HIP_BACKWARD"]
        else:
            os.environ["GAR_HIP_BACKWARD"] = old
        if old_spd is None:
            os.environ.pop("GAR_HIP_SPD_ACCEPT", None)
        else:
            os.environ["GAR_HIP_SPD_ACCEPT"] = old_spd


def check_bulk_gains(prob, mueq, lib_path=None, num_legs=1, devices=None):
    """gar_hip_fetch_results / gar_hip_get_gains_all against the per-stage gar_hip_get_gains and the
    per-part gar_hip_get_solution: bitwise the same numbers, one copy instead of 3 (N+1) + 4."""
    dims = [k.dims for k in prob.stages]
    s = BatchedRiccatiSolver(dims, prob.nc0, batch=2, num_legs=num_legs, lib_path=lib_path, devices=devices)
    s.upload([prob, prob])
    assert s.backward(mueq) and s.forward()
    if num_legs > 1:
        s.collapse_feedback()
    for b in (1, 0):
        ffs, fbs = s.gains_all(b)
        for t in range(prob.horizon + 1):
            f = s.factor(t, b)
            assert np.array_equal(ffs[t], f.ff) and np.array_equal(fbs[t], f.fb), t
        rec, _, _ = s.fetch_results(b, gains=False)
        flat = np.concatenate([np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0)
                               for part in s.solution(b)])
        assert np.array_equal(rec[:flat.size], flat)     # (padded solvers too: the record is the caller's)
    return s


def check_cycle_append_ring(lib_path=None, nx=8, nu=4, horz=5, cycles=8, family=None, dense=False, pipeline=False):
    """MPC cycling as a ring (proximal-riccati.hxx:79-86, tests/mpc-cycle.cpp): after cycleAppend only the
    NEW last-but-one knot is uploaded -- every other knot must still be where the kernels look for it,
    through more cycles than there are stages (the ring wraps) -- and the sweep must match the oracle on
    the caller's rotated problem: solution, every stage's gains and value function."""
    import os
    old = os.environ.get("GAR_HIP_BACKWARD")
    if family:
        os.environ["GAR_HIP_BACKWARD"] = family
    try:
        rng = np.random.default_rng(77)
        probs = [synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W") for _ in range(2)]
        dims = [k.dims for k in probs[0].stages]
        s = BatchedRiccatiSolver(dims, probs[0].nc0, batch=2, lib_path=lib_path, dense=dense)
        s.upload(probs)
        mu = 1e-10
        if pipeline:   # the pipelined schedule (gar_hip_set_pipeline): the ring offset reaches gar_forward_lean too
            s.set_pipeline(2)
            plain_backward, plain_forward = s.backward, s.forward

            def _bw(m):
                s.backward_async(m)
                return True

            def _fw():
                s.forward_async()
                s.sync()
                return s.num_failed() == 0
            s.backward, s.forward = _bw, _fw
        assert s.backward(mu) and s.forward()
        for c in range(cycles):
            s.cycle_append(dims[0])
            for b, p in enumerate(probs):
                new = synth.generate_knot(rng, nx, nu, mode="W")
                p.stages[:horz] = p.stages[1:horz] + [new]        # the caller's own rotation
                s.upload_knot(b, horz - 1, new)                   # ONLY the new knot goes to the device
            assert s.backward(mu) and s.forward()
            for b, p in enumerate(probs):
                if dense:
                    _, _, ref = oracle_serial(p, mu)
                else:
                    _, osol, ref = oracle_serial(p, mu)
                for A, B in zip(s.solution(b), ref):
                    assert maxdiff(A, B) <= 1e-9 * scale_of(ref), (c, b)
                if not dense:
                    class D:
                        def __getitem__(self, t, b=b):
                            return s.factor(t, b)
                    compare_factors(D(), osol, horz, 1e-9, names=("ff", "fb"), vnames=("Vxx", "vx"))
        return s
    finally:
        if family:
            if old is None:
                del os.environ["GAR_HIP_BACKWARD"]
            else:
                os.environ["GAR_HIP_BACKWARD"] = old


def check_constrained_legs_fold(lib_path=None, shapes=((8, 4, 4, 11, 3, 1e-6), (16, 8, 8, 9, 2, 1e-7)), tol=1e-8):
    """Equality-constrained knots in leg mode (the reference's BM_parallel configuration, bench/gar-riccati.cpp:64-90)
    on the UNCONSTRAINED wave-leg kernels through the fold of gar_fold.hpp (Q += C^T C / mu, q += C^T d / mu when
    D = 0; Z, zff from C, d): solution, every stage's ff = [kff; zff; yff] / fb = [K; Z; Aff] / fth / value function,
    collapsed K0 against the oracle's leg-parallel solver -- with a dense C, with constraints on SOME knots only --
    and problems with D != 0, which the same solver hands to the generic leg kernels: one batch mixes both kinds."""
    from aligator_amd.gar import BatchedRiccatiSolver
    rng = np.random.default_rng(41)
    for (nx, nu, nc, horz, legs, mu) in shapes:
        prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
        for k in prob.stages:
            k.C[...] = rng.uniform(-1, 1, k.C.shape)
        # (multipliers of order 1/mu: v and lambda are judged against what the problem's conditioning allows)
        par = check_parallel(prob, mu, legs, tol, lib_path, conditioned=True)
        assert par._impl.kernel_name.startswith(f"wave_leg<{nx},{nu}>+fold")
        coupled = prob.copy()
        for k in coupled.stages[1:horz:3]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
        par = check_parallel(coupled, mu, legs, tol, lib_path, conditioned=True)   # (round 6: the constrained segment legs underneath)
        os.environ["GAR_HIP_CSTR_SEG_LEGS"] = "0"                                  # ... and the any-dimension leg kernels, as before
        try:
            par = check_parallel(coupled, mu, legs, tol, lib_path, conditioned=True)
            assert par._impl.kernel_name == f"wave_leg<{nx},{nu}>+fold"
        finally:
            del os.environ["GAR_HIP_CSTR_SEG_LEGS"]
        # one batch, both kinds: problem 1 has D != 0
        probs = [prob, coupled, prob.copy()]
        probs[2].stages[0].q[...] += 1.0
        dims = [k.dims for k in prob.stages]
        s = BatchedRiccatiSolver(dims, prob.nc0, batch=3, num_legs=legs, lib_path=lib_path)
        s.upload(probs)
        assert s.backward(mu) and s.forward()
        for b, p in enumerate(probs):
            _, osol, ref = oracle_serial(p, mu)
            sc = scale_of(ref)
            for A, B in zip(s.solution(b), ref):
                assert maxdiff(A, B) <= max(tol, 1e-7) * sc, b
            opar = ora.ParallelRiccatiSolver(to_oracle(p), legs)
            opar.backward(mu)

            class D:
                def __getitem__(self, t, b=b):
                    return s.factor(t, b)
            compare_factors(D(), opar, horz, max(tol, 1e-7))
        ffs, fbs = s.gains_all(1)                                         # bulk read-back of a generic-family problem
        for t in range(horz + 1):
            assert np.array_equal(fbs[t], s.factor(t, 1).fb) and np.array_equal(ffs[t], s.factor(t, 1).ff)
    # constraints on SOME knots only (mixed nc): still one uniform (nx, nu), still folded
    nx, nu, horz = 8, 4, 10
    knots = []
    for t in range(horz):
        knots.append(synth.generate_knot(rng, nx, nu, nc=(3 if t % 3 == 1 else 0), mode="W"))
    knots.append(synth.generate_knot(rng, nx, 0, nc=2, mode="W"))
    from aligator_amd.lqr import LqrProblem
    mixed = LqrProblem(knots, nx)
    mixed.G0[...] = -np.eye(nx)
    mixed.g0[...] = rng.standard_normal(nx)
    par = check_parallel(mixed, 1e-6, 3, tol, lib_path, conditioned=True)
    assert par._impl.kernel_name.startswith(f"wave_leg<{nx},{nu}>+fold")

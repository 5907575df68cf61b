"""Kernel-logic tests on the wave emulator (tests/emu): the UNMODIFIED kernel
and C-ABI sources of aligator_amd/csrc run on host threads (one per lane), so
index maps, LDS plans and barrier placement are checked on CPU.  This is test
infrastructure only: the product never loads the emulator build, and these
tests prove nothing about GPU execution (tests/test_gpu_parity.py does)."""
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import synth
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)


@pytest.mark.parametrize("horz", [4, 8, 16])
def test_riccati_short_horz_pb(horz):                         # tests/gar/riccati.cpp:26-85
    pc.check_serial(synth.short_horizon_problem(horz), 1e-14, 1e-9, EMU, kkt_tol=1e-9)


def test_riccati_one_knot_prob():                             # riccati.cpp:87-105
    prob = synth.generate_lq_problem(1, np.zeros(2), 0, 2, 2)
    pc.check_serial(prob, 1e-13, 1e-10, EMU, kkt_tol=1e-10)


@pytest.mark.parametrize("mode", ["F", "W"])
def test_riccati_random_large_problem(mode):                  # riccati.cpp:107-139 (short horizon)
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(42, np.zeros(nx), 6, nx, nu, mode=mode)
    pc.check_serial(prob, 1e-14, pc.TOL[mode], EMU, kkt_tol=1e-6 if mode == "F" else 1e-9)


def test_riccati_parametric():                                # riccati.cpp:157-192
    rng = np.random.default_rng(9)
    nx, nu, nth = 10, 4, 1
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 12, nx, nu, nth=nth)
    theta = rng.uniform(-1, 1, nth)
    solver, sol, _ = pc.check_serial(prob, 1e-12, 1e-9, EMU, theta=theta, kkt_tol=1e-9)
    for arr in (solver.kkt0.ff, solver.kkt0.fth, solver.thGrad, solver.thHess,
                solver.datas[0].vm.vt, solver.datas[0].vm.Vxt, solver.datas[0].vm.Vtt):
        assert np.isfinite(arr).all()


def test_constrained_stages_and_2x2_pivots():
    """nc > 0 at every stage, tiny mu: the reduced KKT is indefinite and the
    device Bunch-Kaufman takes 2x2 pivots (riccati-kernel.hxx:232-241)."""
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(6), 5, 6, 3, nc=4, mode="W")
    pc.check_serial(prob, 1e-9, 1e-9, EMU)


def test_parametric_and_constrained():
    rng = np.random.default_rng(6)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(6), 4, 6, 2, nth=3, nc=3, mode="F")
    pc.check_serial(prob, 1e-9, 1e-8, EMU, theta=rng.uniform(-1, 1, 3))


def test_heterogeneous_dimensions():
    """nx, nu vary along the horizon (A is nx2 x nx, lqr-problem.hxx:28-72)."""
    from aligator_amd.lqr import LqrProblem
    rng = np.random.default_rng(8)
    shapes = [(3, 2, 4), (4, 1, 5), (5, 3, 2), (2, 2, 2)]     # (nx, nu, nx2)
    knots = [synth.generate_knot(rng, nx, nu, nx2=nx2, mode="W") for nx, nu, nx2 in shapes]
    knots.append(synth.generate_knot(rng, 2, 0, mode="W"))
    prob = LqrProblem(knots, 3)
    prob.G0[...] = -np.eye(3)
    prob.g0[...] = rng.standard_normal(3)
    pc.check_serial(prob, 1e-10, 1e-9, EMU, kkt_tol=1e-9)


@pytest.mark.parametrize("pad,nthreads,horz,nx,nu", [("1", 2, 11, 4, 2), ("1", 4, 13, 6, 3), ("1", 3, 8, 12, 6),
                                                     ("0", 2, 11, 4, 2), ("0", 4, 13, 6, 3)])
def test_parallel_solver_class(monkeypatch, pad, nthreads, horz, nx, nu):       # tests/gar/parallel.cpp:185-245
    # pad = 1: these shapes are padded onto the (8,4) / (12,8) kernels; pad = 0: the generic kernels
    monkeypatch.setenv("GAR_HIP_PAD", pad)
    rng = np.random.default_rng(17)
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horz, nx, nu)
    pc.check_parallel(prob, 1e-9, nthreads, 1e-7, EMU, rounds=1, rng=rng)


@pytest.mark.parametrize("nx,nu,nc,nth,horz,mu", [(6, 3, 0, 0, 6, 1e-12), (8, 4, 3, 0, 4, 1e-6),
                                                  (5, 2, 2, 3, 5, 1e-6), (12, 5, 0, 0, 3, 1e-10)])
def test_dense_solver(nx, nu, nc, nth, horz, mu):               # tests/gar/riccati.cpp:141-155
    """RiccatiSolverDense (csrc/gar_dense.hpp): unconstrained, constrained (terminal knot included),
    parameterised with theta in the forward pass."""
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nth=nth, nc=nc, mode="W")
    theta = rng.standard_normal(nth) if nth else None
    pc.check_dense(prob, mu, 1e-9, EMU, theta=theta, kkt_tol=1e-8)


def test_dense_solver_cycle_append_and_batch():
    """cycleAppend (dense-riccati.hxx:118-146) keeps the solver usable on the rotated problem; the
    batch axis gives every problem its own solve."""
    from aligator_amd.gar import BatchedRiccatiSolver, RiccatiSolverDense
    nx, nu, horz = 6, 3, 5
    rng = np.random.default_rng(2)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    s = RiccatiSolverDense(prob, lib_path=EMU)
    assert s.backward(1e-10)
    new = synth.generate_lq_problem(rng, np.zeros(nx), 1, nx, nu, mode="W").stages[0]
    s.cycleAppend(new)
    prob.stages[:horz - 1] = prob.stages[1:horz]
    prob.stages[horz - 1] = new
    pc.check_dense(prob, 1e-10, 1e-9, EMU, kkt_tol=1e-9)
    assert s.backward(1e-10)
    sol = pc.lqrInitializeSolution(prob)
    s.forward(*sol)
    assert max(pc.lqrComputeKktError(prob, *sol, mueq=1e-10)) <= 1e-9
    probs = [synth.generate_lq_problem(40 + i, np.ones(nx), horz, nx, nu, mode="W") for i in range(3)]
    bs = BatchedRiccatiSolver([k.dims for k in probs[0].stages], probs[0].nc0, batch=3, lib_path=EMU, dense=True)
    bs.upload(probs)
    assert bs.backward(1e-10) and bs.forward()
    for b, p in enumerate(probs):
        _, _, ref = pc.oracle_serial(p, 1e-10)
        for A, B in zip(bs.solution(b), ref):
            assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)


@pytest.mark.parametrize("nx,nu,horz,legs,kernel", [(10, 3, 5, 2, "12,4"), (13, 5, 4, 2, "16,8"), (7, 2, 9, 1, "8,4")])
def test_padded_states_and_controls(nx, nu, horz, legs, kernel):
    """Shapes that are not compiled in run on the next larger specialised kernel: the Python mirror
    adds pinned dummy states (Q = I, A = 0, rows [0 -I] of G0) and dummy controls (R = I, B = 0) and
    strips them from every result, so the caller sees the reference's dimensions (gar.py::_pad_knot)."""
    rng = np.random.default_rng(nx * 7 + nu)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    s, _, _ = pc.check_serial(prob, 1e-10, 1e-9, EMU, kkt_tol=1e-9)
    assert kernel in s._impl.kernel_name and tuple(s._impl.dims[0][:2]) == (nx, nu) and s._impl.padded
    if legs > 1:
        par = pc.check_parallel(prob, 1e-10, legs, 1e-8, EMU, rounds=1, rng=rng)
        assert par._impl.kernel_name.startswith("wave_leg<")


@pytest.mark.parametrize("nthreads,horz,nx,nu", [(3, 11, 8, 4), (2, 7, 12, 4), (2, 5, 16, 8)])
def test_parallel_wave_leg_kernels(nthreads, horz, nx, nu):
    """Uniform unconstrained shapes in leg mode run the one-wave-per-(problem, leg) kernels
    (csrc/gar_wave_leg.hpp): parameterised recursion, leg-end knot, tuples, leg roll-out."""
    from aligator_amd.gar import ParallelRiccatiSolver
    rng = np.random.default_rng(23)
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horz, nx, nu, mode="W")
    par = pc.check_parallel(prob, 1e-10, nthreads, 1e-9, EMU, rounds=1, rng=rng)
    assert par._impl.kernel_name.startswith("wave_leg<")


def test_parallel_wave_leg_batched():
    probs = [synth.generate_lq_problem(300 + i, np.zeros(8), 9, 8, 4, mode="W") for i in range(3)]
    s = pc.check_batched(probs, 1e-10, 1e-9, EMU, num_legs=3)
    assert s.kernel_name.startswith("wave_leg<")


def _leg_solution(probs, legs, mueq, refine=None, threshold=1e-10, backward_ok=None):
    from aligator_amd.gar import BatchedRiccatiSolver
    dims = [k.dims for k in probs[0].stages]
    s = BatchedRiccatiSolver(dims, probs[0].nc0, batch=len(probs), num_legs=legs, lib_path=EMU)
    if refine is not None:
        s.set_refinement(threshold, refine, backward_ok)
    s.upload(probs)
    assert s.backward(mueq) and s.forward()
    return s, [s.solution(b) for b in range(len(probs))]


@pytest.mark.parametrize("legs,horz,nx,nu", [(6, 17, 8, 4), (4, 7, 12, 4)])
def test_condensed_cyclic_reduction_vs_chain_vs_generic(monkeypatch, legs, horz, nx, nu):
    """Three solvers of the leg-boundary system -- block cyclic reduction (csrc/gar_cyclic.hpp, the
    default, here WITHOUT its fallback: refinement off), the wave-scope elimination chain and the
    generic workgroup kernel that follows the reference's block-tridiagonal routine line by line --
    agree with each other and with the serial oracle."""
    probs = [synth.generate_lq_problem(900 + i, np.random.default_rng(i).standard_normal(nx), horz,
                                       nx, nu, mode="W") for i in range(1)]
    s, cyc = _leg_solution(probs, legs, 1e-10, refine=0)
    assert s.kernel_name.startswith("wave_leg<")
    resid, steps = s.condensed_info(0)
    assert resid < 1e-9 and steps == 0
    monkeypatch.setenv("GAR_HIP_CONDENSED", "chain")
    _, chain = _leg_solution(probs, legs, 1e-10)
    monkeypatch.setenv("GAR_HIP_CONDENSED", "generic")
    _, gen = _leg_solution(probs, legs, 1e-10)
    monkeypatch.delenv("GAR_HIP_CONDENSED")
    monkeypatch.setenv("GAR_HIP_LEGS", "generic")
    s4, allgen = _leg_solution(probs, legs, 1e-10)
    assert s4.kernel_name == "generic"
    for b, prob in enumerate(probs):
        _, _, ref = pc.oracle_serial(prob, 1e-10)
        sc = pc.scale_of(ref)
        for sol in (cyc[b], chain[b], gen[b], allgen[b]):
            for A, B in zip(sol, ref):
                assert pc.maxdiff(A, B) <= 1e-9 * sc


def test_condensed_cyclic_fallback_to_chain():
    """A threshold the cyclic-reduction residual cannot meet gates the elimination-chain kernel
    in: it re-solves with the reference's iterative refinement and reports its steps."""
    probs = [synth.generate_lq_problem(950, np.ones(8), 21, 8, 4, mode="W")]
    s, sol = _leg_solution(probs, 4, 1e-10, refine=3, threshold=1e-300, backward_ok=0.0)
    resid, steps = s.condensed_info(0)
    assert steps == 3                       # the chain kernel ran (cyclic reduction reports 0 steps)
    # ... while the default gate lets the cyclic-reduction solve stand on its componentwise backward error
    s2, sol2 = _leg_solution(probs, 4, 1e-10, refine=3, threshold=1e-300)
    assert s2.condensed_info(0)[1] == 0 and 0.0 < s2.condensed_backward_error(0) <= 1e-12
    for A, B in zip(sol2[0], sol[0]):
        assert pc.maxdiff(A, B) <= 1e-10 * pc.scale_of(sol[0])
    _, _, ref = pc.oracle_serial(probs[0], 1e-10)
    for A, B in zip(sol[0], ref):
        assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)


def test_condensed_block_inverse_bunch_kaufman_fallback():
    pc.check_condensed_block_inverse_fallback(EMU)


@pytest.mark.parametrize("leg_waves", ["2", "1"])
def test_leg_kernels_bunch_kaufman_fallback_and_failure(monkeypatch, leg_waves):
    monkeypatch.setenv("GAR_HIP_LEG_WAVES", leg_waves)
    pc.check_leg_kernels_bunch_kaufman_fallback(EMU)


@pytest.mark.parametrize("nc0", [0, 3])
def test_leg_kernels_partial_initial_constraint(nc0):
    """G0 with fewer rows than states (nc0 < nx): block 0 of the condensed system is padded."""
    from aligator_amd.lqr import LqrProblem
    rng = np.random.default_rng(5)
    p0 = synth.generate_lq_problem(rng, rng.standard_normal(8), 13, 8, 4, mode="W")
    prob = LqrProblem(p0.stages, nc0)
    prob.G0[...] = rng.standard_normal((nc0, 8))
    prob.g0[...] = rng.standard_normal(nc0)
    s, sol = _leg_solution([prob], 4, 1e-10, refine=0)
    assert s.kernel_name.startswith("wave_leg<")
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    for A, B in zip(sol[0], ref):
        assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)


def test_parallel_rejects_single_thread():                    # parallel-solver.hxx:42-46
    from aligator_amd.gar import ParallelRiccatiSolver
    prob = synth.generate_lq_problem(1, np.zeros(2), 4, 2, 2)
    with pytest.raises(RuntimeError):
        ParallelRiccatiSolver(prob, 1, lib_path=EMU)


def test_batched_problems():
    probs = [synth.generate_lq_problem(100 + i, np.zeros(5), 6, 5, 2, mode="W") for i in range(3)]
    pc.check_batched(probs, 1e-10, 1e-9, EMU)
    pc.check_batched(probs, 1e-10, 1e-8, EMU, num_legs=2)


def test_failed_factorisation_raises():
    """An exactly-zero pivot column makes BunchKaufman report NumericalIssue; the
    reference throws (riccati-kernel.hxx:239-241), so does the host mirror."""
    from aligator_amd.gar import ProximalRiccatiSolver
    prob = synth.generate_lq_problem(3, np.zeros(3), 3, 3, 2, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    with pytest.raises(RuntimeError, match="LDL"):
        ProximalRiccatiSolver(prob, lib_path=EMU).backward(1e-10)


def test_cycle_append_rotates_factors():                      # proximal-riccati.hxx:79-86
    from aligator_amd.gar import ProximalRiccatiSolver
    prob = synth.generate_lq_problem(21, np.zeros(4), 6, 4, 2, mode="W")
    s = ProximalRiccatiSolver(prob, lib_path=EMU)
    s.backward(1e-10)
    before = [s.getFeedback(t).copy() for t in range(7)]
    s.cycleAppend(prob.stages[5])
    for t in range(5):
        assert np.array_equal(s.getFeedback(t), before[t + 1])
    assert np.array_equal(s.getFeedback(5), np.zeros_like(before[5]))   # re-created factor
    assert np.array_equal(s.getFeedback(6), before[6])                  # terminal factor kept


# ---- the specialised backward kernels: one wave per problem (csrc/gar_wave.hpp,
# the default) and the 4-wave workgroup kernel (csrc/gar_mfma.hpp, GAR_HIP_BACKWARD=wg4)
FAMILIES = [("wave", "wave"), ("wg4", "mfma")]


@pytest.fixture(params=FAMILIES, ids=[f[0] for f in FAMILIES])
def family(request, monkeypatch):
    monkeypatch.setenv("GAR_HIP_BACKWARD", request.param[0])
    return request.param[1]


@pytest.mark.parametrize("nx,nu,horz,mode", [(8, 4, 3, "W"), (12, 4, 5, "W"), (16, 8, 4, "F"),
                                             (36, 12, 4, "W"), (36, 12, 3, "F"), (32, 12, 3, "W")])
def test_mfma_kernel_matches_oracle(nx, nu, horz, mode, family):
    prob = synth.generate_lq_problem(7, np.zeros(nx), horz, nx, nu, mode=mode)
    solver, _, _ = pc.check_serial(prob, 1e-12, pc.TOL[mode], EMU,
                                   kkt_tol=1e-6 if mode == "F" else 1e-9)
    assert solver.kernel_name == f"{family}<{nx},{nu}>"


def test_wave_kernel_odd_and_even_horizons(monkeypatch):
    """The stage body is instantiated twice (register ping-pong): both parities of the
    horizon, and horizon 1, must take the right tail."""
    monkeypatch.setenv("GAR_HIP_BACKWARD", "wave")
    for horz in (1, 2, 5):
        prob = synth.generate_lq_problem(70 + horz, np.zeros(12), horz, 12, 4, mode="W")
        solver, _, _ = pc.check_serial(prob, 1e-12, 1e-9, EMU, kkt_tol=1e-9)
        assert solver.kernel_name == "wave<12,4>"


def test_mfma_kernel_bunch_kaufman_fallback(family):
    """A stage whose Rhat makes Bunch-Kaufman interchange (rule at
    bunchkaufman.hpp:61-83) must take the generic device BK, uniformly for the
    workgroup, and still match the oracle."""
    from oracle import oracle as ora
    nx, nu = 8, 4
    prob = synth.generate_lq_problem(11, np.zeros(nx), 4, nx, nu, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = np.array([[1e-3, 2.0, 0.1, 0.0], [2.0, 1e-3, 0.0, 0.1],
                             [0.1, 0.0, 3.0, 0.2], [0.0, 0.1, 0.2, 4.0]])
        k.B[...] *= 1e-2
    op, osol, _ = pc.oracle_serial(prob, 1e-12)
    pivots = [ora.BunchKaufman(osol.datas(t).Rhat).pivots for t in range(4)]
    assert any(not np.array_equal(p, np.arange(nu)) for p in pivots), "test must force a pivot"
    solver, _, _ = pc.check_serial(prob, 1e-12, 1e-9, EMU)
    assert solver.kernel_name == f"{family}<8,4>"


def test_mfma_kernel_failed_factorisation_raises(family):
    from aligator_amd.gar import ProximalRiccatiSolver
    prob = synth.generate_lq_problem(3, np.zeros(8), 3, 8, 4, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    s = ProximalRiccatiSolver(prob, lib_path=EMU)
    assert s.kernel_name == f"{family}<8,4>"
    with pytest.raises(RuntimeError, match="LDL"):
        s.backward(1e-10)


def test_mfma_and_generic_kernels_agree(monkeypatch, family):
    prob = synth.generate_lq_problem(5, np.zeros(12), 6, 12, 4, mode="W")
    from aligator_amd.gar import ProximalRiccatiSolver, lqrInitializeSolution
    a = ProximalRiccatiSolver(prob, lib_path=EMU)
    a.backward(1e-12)
    sa = lqrInitializeSolution(prob)
    a.forward(*sa)
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    g = ProximalRiccatiSolver(prob, lib_path=EMU)
    assert g.kernel_name == "generic" and a.kernel_name == f"{family}<12,4>"
    g.backward(1e-12)
    sg = lqrInitializeSolution(prob)
    g.forward(*sg)
    for A, B in zip(sa, sg):
        assert pc.maxdiff(A, B) <= 1e-11


# ---- edge cases of the boundary -------------------------------------------------
@pytest.mark.parametrize("nc0", [0, 2])
def test_partial_or_no_initial_constraint(nc0):
    """G0 with fewer rows than nx (or none): kkt0 = [Vxx0 G0^T; G0 0] is (nx+nc0)-dimensional
    (proximal-riccati.hxx:42-60)."""
    from aligator_amd.lqr import LqrProblem
    rng = np.random.default_rng(40 + nc0)
    nx, nu = 4, 2
    knots = [synth.generate_knot(rng, nx, nu, mode="W") for _ in range(5)]
    knots.append(synth.generate_knot(rng, nx, 0, mode="W"))
    prob = LqrProblem(knots, nc0)
    if nc0:
        prob.G0[...] = rng.standard_normal((nc0, nx))
        prob.g0[...] = rng.standard_normal(nc0)
    pc.check_serial(prob, 1e-10, 1e-9, EMU, kkt_tol=1e-9)


def test_dimensions_beyond_one_cu_are_refused():
    """A stage whose working set exceeds a CU's 160 KiB of LDS is refused at creation with
    GAR_HIP_ERR_UNSUPPORTED (no silent fallback)."""
    from aligator_amd.gar import BatchedRiccatiSolver
    dims = [(160, 40, 0, 160, 0)] * 2 + [(160, 0, 0, 160, 0)]
    with pytest.raises(RuntimeError, match="LDS"):
        BatchedRiccatiSolver(dims, 160, batch=1, lib_path=EMU)


def test_bad_arguments_are_rejected():
    from aligator_amd.gar import BatchedRiccatiSolver
    with pytest.raises(RuntimeError):
        BatchedRiccatiSolver([(4, 2, 0, 4, 0), (4, 0, 0, 4, 0)], 4, batch=0, lib_path=EMU)
    s = BatchedRiccatiSolver([(4, 2, 0, 4, 0), (4, 0, 0, 4, 0)], 4, batch=2, lib_path=EMU)
    with pytest.raises(RuntimeError, match="out of range"):
        s._check(s._L.gar_hip_get_gains(s.handle, 0, 5, None, None, None))
    with pytest.raises(RuntimeError, match="out of range"):
        s._check(s._L.gar_hip_get_gains(s.handle, 7, 0, None, None, None))


def test_batch_of_identical_and_distinct_problems_agree_with_singles():
    """Batch axis: problem b of a batch equals the same problem solved alone (bitwise)."""
    from aligator_amd.gar import BatchedRiccatiSolver
    probs = [synth.generate_lq_problem(300 + i, np.zeros(8), 5, 8, 4, mode="W") for i in range(3)]
    dims = [k.dims for k in probs[0].stages]
    sb = BatchedRiccatiSolver(dims, 8, batch=3, lib_path=EMU)
    sb.upload(probs)
    sb.backward(1e-12)
    sb.forward()
    for b, p in enumerate(probs):
        s1 = BatchedRiccatiSolver(dims, 8, batch=1, lib_path=EMU)
        s1.upload([p])
        s1.backward(1e-12)
        s1.forward()
        for A, B in zip(sb.solution(b), s1.solution(0)):
            assert pc.maxdiff(A, B) == 0.0


# ---- constrained stages on the wave kernels (csrc/gar_wave.hpp, NC > 0) ---------------------------
@pytest.mark.parametrize("nx,nu,nc,horz,mu", [(8, 4, 4, 7, 1e-6), (16, 8, 8, 4, 1e-9), (8, 4, 4, 1, 1e-3)])
def test_constrained_wave_kernels(nx, nu, nc, horz, mu):
    """Uniform problems with nc constraints on every knot (the reference's bench/gar-riccati.cpp
    structure): reduced KKT system by wave-scope Bunch-Kaufman, Vxx += C^T Z, vs in the forward
    sweep.  Factors (ff = [kff; zff; yff], fb = [K; Z; Aff]), value functions, kkt0 and the
    solution against the oracle."""
    rng = np.random.default_rng(300 + nx + nc)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    solver, sol, ref = pc.check_serial(prob, mu, 1e-8, EMU)
    assert solver.kernel_name == f"wave<{nx},{nu},{nc}>"
    assert len(sol[2]) == horz + 1 and sol[2][horz].size == nc   # vs on every knot, terminal included


def test_constrained_wave_kernels_bunch_kaufman_pivoting():
    pc.check_constrained_pivoting(EMU)


def test_parallel_solver_on_the_reference_bench_shape_nc32():
    """bench/gar-riccati.cpp:64-90 (BM_parallel): nx = 36, nu = 12, nc = 32 in leg mode.  The generic leg
    kernels' LDS plan with both generations of the parameter blocks needs 183 KB; the lean plan (one
    generation in LDS, Vxt', Vtt', vt' read back from stage t+1's record) fits."""
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 5, nx, nu, nc=nc, mode="W")
    par = pc.check_parallel(prob, 1e-8, 2, 1e-7, EMU)
    assert par._impl.kernel_name.startswith("wave_leg<36,12>+fold")      # round 3: folded onto the wave-leg family (gar_fold.hpp)


def test_parallel_solver_nc32_on_the_generic_leg_kernels(monkeypatch):
    """... and the generic leg kernels (the fallback for D != 0) on the same shape: GAR_HIP_FOLD=0."""
    monkeypatch.setenv("GAR_HIP_FOLD", "0")
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 5, nx, nu, nc=nc, mode="W")
    par = pc.check_parallel(prob, 1e-8, 2, 1e-7, EMU)
    assert par._impl.kernel_name == "generic"


def test_constrained_legs_fold_onto_the_wave_leg_kernels():
    pc.check_constrained_legs_fold(EMU)


def test_coupled_constraints_in_leg_mode_on_the_constrained_segment_legs():
    """Round 6 (gar_cstr_seg.hpp): D != 0 in leg mode on the serial constrained chain's stage kernels, leg by leg, + the
    parameter recursion; the chain's three kernels all reached inside legs."""
    pc.check_constrained_legs_segments(EMU)


def test_mueq_zero_on_a_serial_constrained_solver_is_a_reported_failure():
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(8), 6, 8, 4, nc=4, mode="W")
    from aligator_amd.gar import ProximalRiccatiSolver
    s = ProximalRiccatiSolver(prob, lib_path=EMU)
    assert s.kernel_name == "wave<8,4,4>"
    with pytest.raises(RuntimeError, match="Failed stage LDL factorization"):
        s.backward(0.0)
    assert s.backward(1e-6)


def test_constrained_wave_kernels_decoupled_dense_c_and_alternating_d():
    pc.check_constrained_decoupled(EMU, shapes=((8, 4, 4, 5, 1e-6), (16, 8, 8, 4, 1e-8), (36, 12, 32, 3, 1e-9)))



def test_device_written_knots_survive_host_set_init():
    """A device-resident producer writes the knots in place (gar_hip_device_problems); a later
    host-side set_init / upload_knot must flush ONLY what the host wrote (the whole
    staging buffer used to be copied over the device records)."""
    import ctypes
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N = 8, 4, 6
    probs = [synth.generate_lq_problem(50 + i, np.ones(nx), N, nx, nu, mode="W") for i in range(2)]
    dims = [k.dims for k in probs[0].stages]
    s = BatchedRiccatiSolver(dims, nx, batch=2, lib_path=EMU)
    packed = np.concatenate([s.pack_device(p) for p in probs])                  # the DEVICE's record format
    ctypes.memmove(s.device_pointers()[0], packed.ctypes.data, packed.nbytes)   # "device" write in place
    x0 = np.full(nx, 0.5)
    for b, p in enumerate(probs):
        p.g0[...] = x0
        s.set_init(b, p.G0, p.g0)                       # host write: G0 | g0 only
    k = probs[1].stages[3]
    k.q[...] += 1.0
    s.upload_knot(1, 3, k)                              # host write: one knot of problem 1
    assert s.backward(1e-10) and s.forward()
    for b, p in enumerate(probs):
        _, _, ref = pc.oracle_serial(p, 1e-10)
        for A, B in zip(s.solution(b), ref):
            assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)
    # and the device records still hold the producer's knots
    back = s.download_packed(0, 2)
    packed2 = np.concatenate([s.pack(p) for p in probs])
    assert np.array_equal(back, packed2)


def test_wave_kernel_second_bunch_kaufman_test():
    pc.check_second_bunch_kaufman_test(EMU)


@pytest.mark.parametrize("nx,nu,horz,legs", [(8, 4, 7, 1), (12, 4, 9, 3), (5, 2, 6, 1), (10, 3, 8, 1)])
def test_bulk_gains_and_solution_readback(nx, nu, horz, legs):
    """wave / wave-leg kernels (fbT2 device order), generic kernels (row-major), a padded shape."""
    prob = synth.generate_lq_problem(300 + nx, np.ones(nx), horz, nx, nu, mode="W")
    pc.check_bulk_gains(prob, 1e-10, EMU, num_legs=legs)


@pytest.mark.parametrize("nx,nu,family,dense", [(8, 4, "wave", False), (8, 4, "wg4", False), (5, 2, None, False),
                                                (12, 4, "wave", False), (6, 3, None, True)])
def test_cycle_append_is_a_ring(nx, nu, family, dense):
    s = pc.check_cycle_append_ring(EMU, nx=nx, nu=nu, horz=4, cycles=6, family=family, dense=dense)  # (wraps: cycles > horz)
    if family:
        assert s.kernel_name.startswith("wave<" if family == "wave" else "mfma<")


@pytest.mark.parametrize("variant,name", [("pair", "pair<56,24>"), ("single", "wave<56,24>")])
def test_wide_shape_kernels(monkeypatch, variant, name):
    """The Talos-walk LQ shape (bench/talos-walk.cpp:20-28: nx = 56, nu = 22 -> controls padded to 24): two
    waves per problem (gar_wave_pair.hpp, the default) and the one-wave stage generalised to five tile
    columns (GAR_HIP_WIDE=single); fb row-major, generic initial stage and forward sweep.  Generators W and
    F, a stage that makes Bunch-Kaufman pivot (generic device Bunch-Kaufman on the 24 x 24 Rhat), a failure."""
    from aligator_amd.gar import ProximalRiccatiSolver
    monkeypatch.setenv("GAR_HIP_WIDE", variant)
    prob = synth.generate_lq_problem(560, np.ones(56), 3, 56, 22, mode="W")
    solver, _, _ = pc.check_serial(prob, 1e-10, 1e-9, EMU, kkt_tol=1e-9)
    assert solver.kernel_name == name
    probf = synth.generate_lq_problem(561, np.zeros(56), 2, 56, 24, mode="F")
    solver, _, _ = pc.check_serial(probf, 1e-10, 1e-6, EMU)
    assert solver.kernel_name == name
    piv = synth.generate_lq_problem(562, np.zeros(56), 2, 56, 24, mode="W")
    for k in piv.stages[:-1]:
        R = np.eye(24) * 3.0
        R[0, 0] = R[1, 1] = 1e-3
        R[0, 1] = R[1, 0] = 2.0
        k.R[...] = R
        k.S[...] = 0.0
        k.B[...] *= 1e-2
    solver, _, _ = pc.check_serial(piv, 1e-12, 1e-9, EMU)
    assert solver._impl.slow_path_stages()[1] > 0
    bad = synth.generate_lq_problem(563, np.zeros(56), 2, 56, 24, mode="W")
    for k in bad.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    with pytest.raises(RuntimeError, match="LDL"):
        ProximalRiccatiSolver(bad, lib_path=EMU).backward(1e-10)


@pytest.mark.parametrize("nx,nu,horz,mode", [(36, 12, 4, "W"), (36, 12, 3, "F"), (12, 4, 5, "W"), (8, 4, 4, "W")])
def test_pair_kernel_on_the_narrow_shapes(monkeypatch, nx, nu, horz, mode):
    """GAR_HIP_BACKWARD=pair: the two-waves-per-problem sweep (gar_wave_pair.hpp) on the shapes whose tile
    layout allows the split (fbT2 gains, gar_forward_mfma)."""
    monkeypatch.setenv("GAR_HIP_BACKWARD", "pair")
    prob = synth.generate_lq_problem(80 + nx, np.ones(nx), horz, nx, nu, mode=mode)
    solver, _, _ = pc.check_serial(prob, 1e-12, pc.TOL[mode], EMU, kkt_tol=1e-6 if mode == "F" else 1e-9)
    assert solver.kernel_name == (f"pair<{nx},{nu}>" if nx == 36 else f"wave<{nx},{nu}>")


@pytest.mark.parametrize("nx,nu", [(8, 4), (7, 3)])
def test_rejected_cycle_append_leaves_the_ring_intact(nx, nu):
    """a cycle_append whose new layout is rejected must leave the solver exactly as it was --
    ring position, stage offsets, descriptors, device records -- also AFTER earlier uniform cycles (ring0 != 0, the
    normal MPC case), on a plain and on a padded shape.  The new configuration is validated on a trial object."""
    from aligator_amd.gar import BatchedRiccatiSolver
    horz = 6
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    dims = [k.dims for k in prob.stages]
    s = BatchedRiccatiSolver(dims, prob.nc0, batch=1, lib_path=EMU)
    s.upload([prob])
    assert s.backward(1e-10) and s.forward()
    for _ in range(2):                                             # two uniform cycles: ring0 = 2
        s.cycle_append(dims[0])
        new = synth.generate_knot(rng, nx, nu, mode="W")
        prob.stages[:horz] = prob.stages[1:horz] + [new]
        s.upload_knot(0, horz - 1, new)
    assert s.backward(1e-10) and s.forward()
    before, offs = s.solution(0), s.stage_offsets.copy()
    gains = [s.factor(t).fb.copy() for t in range(horz + 1)]
    for bad in ((-1, nu, 0, nx, 0), (400, 100, 0, 400, 0)):       # invalid dims; a knot no kernel's LDS plan holds
        with pytest.raises(RuntimeError):
            s.cycle_append(bad)
        assert np.array_equal(s.stage_offsets, offs)
        assert s.backward(1e-10) and s.forward()
        after = s.solution(0)
        for A, B in zip(after, before):
            assert pc.maxdiff(A, B) == 0.0
        for t in range(horz + 1):
            assert np.array_equal(s.factor(t).fb, gains[t])
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    for A, B in zip(before, ref):
        assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)
    # ... and a VALID change of dimensions after the ring cycles still works (layout rebuilt, data re-uploaded)
    wider = (nx, nu + 1, 0, nx, 0)
    s.cycle_append(wider)
    prob.stages[:horz] = prob.stages[1:horz] + [synth.generate_knot(rng, nx, nu + 1, mode="W")]
    s.upload([prob])
    assert s.backward(1e-10) and s.forward()
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    for A, B in zip(s.solution(0), ref):
        assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)


def test_wide_shape_in_leg_mode_segment_legs():
    """ParallelRiccatiSolver on the Talos-walk LQ shape (56, 22) -- how the reference benchmarks it
    (bench/talos-walk.cpp:102-127, bench/lqr.cpp:112-134) -- on the segment-leg family (gar_leg_seg.hpp): the
    two-wave stage kernel over each leg's stage range from a zero value function behind the leg end, the parameter
    part (Kth, Yth, Vxt, Vtt, vt) by the generic matrix recursion.  Solution, every stage's factors incl. the
    leg-end records (yff, Aff, Yth zero like terminalSolve leaves them), collapsed K0 against the oracle's
    leg-parallel solver; padded (56, 22) and native (56, 24); a batch."""
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(56), 7, 56, 22, mode="W")
    par = pc.check_parallel(prob, 1e-10, 3, 1e-9, EMU)
    assert par._impl.kernel_name == "pair_leg<56,24>" and par._impl.padded
    probs = [synth.generate_lq_problem(70 + i, np.ones(56), 5, 56, 24, mode="W") for i in range(2)]
    s = pc.check_batched(probs, 1e-10, 1e-9, EMU, num_legs=2)
    assert s.kernel_name == "pair_leg<56,24>" and not s.padded


def test_staged_problem_goes_out_in_few_copies():
    """One Newton iteration's re-read of the problem (gar_hip_backward_blocks) leaves the pinned staging area in
    1 MiB pieces: a knot is ONE dirty range also when Q and R are kept as packed lower triangles (the holes inside
    their blocks travel along) -- counted on the emulator's hipMemcpyAsync: a handful of copies per call, not two
    per knot (round 4 found 2 N + 3 of them behind the packed triangles)."""
    import ctypes as C
    from aligator_amd.gar import BatchedRiccatiSolver
    lib = C.CDLL(EMU)
    lib.emu_memcpy_async_count.restype = C.c_longlong
    for nx, nu, N, legs in ((12, 4, 48, 1), (8, 4, 40, 4), (10, 3, 30, 1)):
        prob = synth.generate_lq_problem(3, np.zeros(nx), N, nx, nu, mode="W")
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, lib_path=EMU)
        assert s.backward_blocks(prob, 1e-10)
        c0 = lib.emu_memcpy_async_count()
        assert s.backward_blocks(prob, 1e-10)
        assert lib.emu_memcpy_async_count() - c0 <= 8, (nx, nu, N)
        s.close()


def test_serial_family_keeps_vxx_as_its_packed_lower_triangle():
    """The record format csrc/gar_layout.h documents (gar_sym_index): in the serial one-wave family the Vxx block of
    a factor record holds the lower triangle, rectangular packed, in its first nx (nx + 1) / 2 doubles and nothing
    else of the block is written -- read here straight from the (emulated) device record and compared with what
    gar_hip_get_value unpacks; every stage incl. the terminal one, and after a cycleAppend (records move whole)."""
    import ctypes as C
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N = 12, 4, 5
    prob = synth.generate_lq_problem(77, np.ones(nx), N, nx, nu, mode="W")
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, lib_path=EMU)
    assert s.kernel_name in ("wave<12,4>", "mfma<12,4>")
    s.upload([prob])
    fac = s.device_pointers()[1]
    raw = (C.c_double * s.device_factors_doubles).from_address(fac)
    np.frombuffer(raw, dtype=np.float64)[:] = -7.0          # sentinel: what the sweep does not write stays
    s.backward(1e-10)
    buf = np.frombuffer(raw, dtype=np.float64)

    def sym_index(n, i, j):                                  # gar_layout.h: gar_sym_index(1, n, i, j)
        a, b = max(i, j), min(i, j)
        return b * (n + 1) + (a - b) if 2 * b < n else (n - 1 - b) * (n + 1) + (b + 1) + (a - b)

    assert sorted(sym_index(nx, i, j) for j in range(nx) for i in range(j, nx)) == list(range(nx * (nx + 1) // 2))
    for t in range(N + 1):
        off = (C.c_int64 * 6)()
        s._check(s._L.gar_hip_stage_offsets(s.handle, t, off))
        nu_t = nu if t < N else 0
        o_vxx = off[1] + (nu_t + nx) + (nu_t + nx) * nx      # ff | fb | Vxx (gar_factor_layout, nc = nth = 0)
        Vxx = np.zeros((nx, nx), order="F")
        vx = np.zeros(nx)
        s._check(s._L.gar_hip_get_value(s.handle, 0, t, Vxx.ctypes.data_as(C.POINTER(C.c_double)),
                                        vx.ctypes.data_as(C.POINTER(C.c_double)), None, None, None))
        assert np.abs(Vxx - Vxx.T).max() == 0.0 and np.abs(Vxx).max() > 0
        block = buf[o_vxx:o_vxx + nx * nx]
        for j in range(nx):
            for i in range(j, nx):
                assert block[sym_index(nx, i, j)] == Vxx[i, j]
        assert (block[nx * (nx + 1) // 2:] == -7.0).all()    # the other half of the block is never touched
    s.close()


@pytest.mark.parametrize("nx,nu,horz,legs", [(10, 4, 12, 2), (10, 4, 13, 3), (20, 7, 23, 5), (9, 3, 16, 8)])
def test_generic_condensed_solve_with_the_leg_states_eliminated_first(monkeypatch, nx, nu, horz, legs):
    """The any-dimension leg path: gar_condensed_leg_eliminate (one factorisation and substitution per leg, all
    legs at once), the chain on the J remaining blocks, gar_condensed_leg_states -- against the oracle's
    leg-parallel solver, and against the same solver with the reduction switched off (the reference's elimination
    order on all 2 J blocks).  The fast result must STAND (gar_hip_condensed_resolved = 0); with a residual
    threshold no solver can meet and the backward-error gate off it must be redone in the reference's order."""
    from aligator_amd.gar import BatchedRiccatiSolver
    monkeypatch.setenv("GAR_HIP_PAD", "0")
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    rng = np.random.default_rng(100 * nx + legs)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    par = pc.check_parallel(prob, 1e-10, legs, 1e-9, EMU)
    assert par._impl.kernel_name == "generic" and not par._impl.condensed_resolved(0)
    dims = [k.dims for k in prob.stages]
    sols = {}
    for reduced in ("1", "0"):
        monkeypatch.setenv("GAR_HIP_CONDENSED_REDUCED", reduced)
        s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, lib_path=EMU)
        s.upload([prob])
        s.backward(1e-10)
        s.forward()
        sols[reduced] = s.solution(0)
        assert not s.condensed_resolved(0)
        if reduced == "1":   # an unreachable threshold, no backward-error pass: the full chain has to redo it
            s.set_refinement(1e-300, 2, backward_ok=0.0)
            s.backward(1e-10)
            s.forward()
            assert s.condensed_resolved(0)
            again = s.solution(0)
            sc = max(1.0, max(float(np.abs(v).max()) for v in again[3]))
            assert max(float(np.abs(a - b).max()) for A, B in zip(again, sols["1"]) for a, b in zip(A, B) if a.size) <= 1e-10 * sc
        s.close()
    sc = max(1.0, max(float(np.abs(v).max()) for v in sols["0"][3]))
    assert max(float(np.abs(a - b).max()) for A, B in zip(sols["1"], sols["0"]) for a, b in zip(A, B) if a.size) <= 1e-10 * sc


@pytest.mark.parametrize("nx,nu,horz,legs,nc0", [(20, 7, 23, 5, None), (9, 3, 16, 8, None), (6, 2, 27, 13, None),
                                                  (10, 4, 19, 4, 3), (10, 4, 14, 7, 0), (12, 5, 34, 16, None),
                                                  (9, 3, 9, 2, None), (9, 3, 10, 3, None)])
def test_generic_condensed_solve_by_block_cyclic_reduction(monkeypatch, nx, nu, horz, legs, nc0):
    """gar_condensed_cr.hpp: with the leg states gone the J remaining blocks are reduced level by level, a workgroup
    per block and level (log2 J dependent steps; any J, not only powers of two; block 0 of dimension nc0 < nx or 0;
    blocks below 8 rows on Bunch-Kaufman).  Against the serial oracle, against the one-workgroup chain on the same
    reduced system (GAR_HIP_CONDENSED_CR=0) and -- with a threshold no solver meets -- redone by the gated full chain."""
    from aligator_amd.gar import BatchedRiccatiSolver
    from aligator_amd.lqr import LqrProblem
    monkeypatch.setenv("GAR_HIP_PAD", "0")
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    rng = np.random.default_rng(1000 * nx + legs)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    if nc0 is not None:
        prob = LqrProblem(prob.stages, nc0)
        prob.G0[...] = rng.standard_normal((nc0, nx))
        prob.g0[...] = rng.standard_normal(nc0)
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    dims = [k.dims for k in prob.stages]
    sols = {}
    for cr in ("1", "0"):
        monkeypatch.setenv("GAR_HIP_CONDENSED_CR", cr)
        s = BatchedRiccatiSolver(dims, prob.nc0, batch=1, num_legs=legs, lib_path=EMU)
        assert s.kernel_name == "generic"
        assert s.condensed_solver_name == ("reduced+cyclic" if cr == "1" else "reduced+chain")
        s.upload([prob])
        assert s.backward(1e-10) and s.forward()
        sols[cr] = s.solution(0)
        assert not s.condensed_resolved(0)
        resid, steps = s.condensed_info(0)
        assert resid <= 1e-9 * pc.scale_of(ref)
        for A, B in zip(sols[cr], ref):
            assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)
        if cr == "1":
            s.set_refinement(1e-300, 2, backward_ok=0.0)
            assert s.backward(1e-10) and s.forward()
            assert s.condensed_resolved(0)
            for A, B in zip(s.solution(0), sols["1"]):
                assert pc.maxdiff(A, B) <= 1e-10 * pc.scale_of(ref)
        s.close()
    for A, B in zip(sols["1"], sols["0"]):
        assert pc.maxdiff(A, B) <= 1e-10 * pc.scale_of(ref)
    monkeypatch.delenv("GAR_HIP_CONDENSED_CR")
    s = BatchedRiccatiSolver(dims, prob.nc0, batch=1, num_legs=legs, lib_path=EMU)   # default: from 4 legs on
    assert s.condensed_solver_name == ("reduced+cyclic" if legs >= 4 else "reduced+chain")
    s.close()
    s = BatchedRiccatiSolver(dims, prob.nc0, batch=1, num_legs=3, lib_path=EMU)
    assert s.condensed_solver_name == "reduced+chain"
    s.close()


def test_generic_condensed_cyclic_reduction_on_a_batch(monkeypatch):
    """A batch through the level kernels (grid (blocks of the level, batch)): every problem against its own oracle."""
    from aligator_amd.gar import BatchedRiccatiSolver
    monkeypatch.setenv("GAR_HIP_PAD", "0")
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    nx, nu, horz, legs = 9, 4, 17, 6
    probs = [synth.generate_lq_problem(300 + i, np.random.default_rng(i).standard_normal(nx), horz, nx, nu, mode="W")
             for i in range(3)]
    s = pc.check_batched(probs, 1e-10, 1e-9, EMU, num_legs=legs)
    assert s.condensed_solver_name == "reduced+cyclic"
    assert not any(s.condensed_resolved(b) for b in range(3))


@pytest.mark.parametrize("nx,nu,nc,horz,legs", [(10, 8, 20, 6, 1), (10, 8, 20, 9, 3), (6, 6, 40, 4, 1)])
def test_generic_constrained_stage_on_the_blocked_bunch_kaufman(monkeypatch, nx, nu, nc, horz, legs):
    """The any-dimension backward kernel with a coupled constrained stage of nu + nc >= 24: its reduced KKT matrix
    [Rhat D^T; D -mu I] goes through wg_bk_factor_blocked (panel by one wave, MFMA trailing updates) on a
    1 024-thread workgroup -- serial and in leg mode, against the oracle (every factor block)."""
    monkeypatch.setenv("GAR_HIP_PAD", "0")
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    rng = np.random.default_rng(7 * nc + legs)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    if legs == 1:
        solver, _, _ = pc.check_serial(prob, 1e-6, 1e-8, EMU, conditioned=True)
        assert solver.kernel_name == "generic"
    else:
        par = pc.check_parallel(prob, 1e-6, legs, 1e-8, EMU, conditioned=True)
        assert par._impl.kernel_name == "generic"


@pytest.mark.parametrize("condensed", ["cyclic", "chain"])
def test_condensed_resolved_is_defined_on_the_specialised_leg_families(monkeypatch, condensed):
    """gar_hip_condensed_resolved on a wave_leg family: the gated wave-scope chain behind block cyclic
    reduction writes the flag -- 0 when the fast result stood (also before any solve: the scratch is zeroed at
    allocation), 1 when an unreachable threshold with the backward-error gate off forces the re-solve in the
    reference's order -- and the re-solved result equals the fast one."""
    from aligator_amd.gar import BatchedRiccatiSolver
    if condensed == "chain":
        monkeypatch.setenv("GAR_HIP_CONDENSED", "chain")
    nx, nu, horz, legs = 8, 4, 13, 3
    prob = synth.generate_lq_problem(11, np.zeros(nx), horz, nx, nu, mode="W")
    s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=2, num_legs=legs, lib_path=EMU)
    assert s.kernel_name == "wave_leg<8,4>" and s.condensed_solver_name == condensed
    assert not s.condensed_resolved(0) and not s.condensed_resolved(1)   # defined before the first solve
    s.upload([prob, prob])
    assert s.backward(1e-10) and s.forward()
    fast = s.solution(1)
    assert not s.condensed_resolved(0) and not s.condensed_resolved(1)
    if condensed == "cyclic":
        s.set_refinement(1e-300, 2, backward_ok=0.0)
        assert s.backward(1e-10) and s.forward()
        assert s.condensed_resolved(0) and s.condensed_resolved(1)
        again = s.solution(1)
        sc = max(1.0, max(float(np.abs(v).max()) for v in again[3]))
        assert max(float(np.abs(a - b).max()) for A, B in zip(again, fast) for a, b in zip(A, B) if a.size) <= 1e-10 * sc


def test_padding_is_dropped_when_no_specialised_kernel_binds(monkeypatch):
    """a leg-mode problem whose (nx, nu) would be padded onto a specialised shape but whose legs the
    specialised leg kernels do not serve (a leg of one knot) must run the any-dimension kernels on ITS OWN
    dimensions, not on the padded ones."""
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, horz, legs = 10, 3, 5, 4                      # 6 knots over 4 legs: legs of one knot
    dims = [(nx, nu, 0, nx, 0)] * horz + [(nx, 0, 0, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, lib_path=EMU)
    assert s.kernel_name == "generic" and not s.padded and tuple(s.device_dims[0][:2]) == (nx, nu)
    s.close()
    serial = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=1, lib_path=EMU)
    assert serial.padded and "<12,4>" in serial.kernel_name   # the serial solver of the same shape IS padded
    serial.close()
    longer = BatchedRiccatiSolver([(nx, nu, 0, nx, 0)] * 11 + [(nx, 0, 0, nx, 0)], nx, batch=1, num_legs=3, lib_path=EMU)
    assert longer.padded and longer.kernel_name == "wave_leg<12,4>"
    longer.close()
    rng = np.random.default_rng(8)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    pc.check_parallel(prob, 1e-10, legs, 1e-8, EMU)


def test_packed_upload_of_constrained_and_parameterised_problems_on_packed_triangle_records():
    """gar_hip_upload_packed on solvers whose records keep Q, R as packed lower triangles (round 4) goes knot by knot:
    constrained knots (C, D, d) and user parameters must travel too (a GPU-only test found them dropped)."""
    from aligator_amd.gar import BatchedRiccatiSolver
    rng = np.random.default_rng(21)
    for nx, nu, nc, nth, horz, mu in ((8, 4, 4, 0, 5, 1e-6), (16, 8, 8, 0, 3, 1e-6)):
        probs = [synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, nth=nth, mode="W") for _ in range(2)]
        s = pc.check_batched(probs, mu, 1e-8, EMU)
        assert s.qr_packed and f"<{nx},{nu},{nc}>" in s.kernel_name
        back = s.download_packed()
        want = np.concatenate([s.pack(p) for p in probs])
        assert np.array_equal(back, want)


def test_backward_blocks_is_the_upload_and_backward_sequence():
    """gar_hip_backward_blocks (the binding's backward() in one ABI call): same bits as gar_hip_upload_stage x (N+1) +
    gar_hip_set_init + gar_hip_backward, serial and leg mode, a padded shape; refused on a batch."""
    from aligator_amd.gar import BatchedRiccatiSolver
    rng = np.random.default_rng(31)
    for nx, nu, horz, legs in ((8, 4, 9, 1), (8, 4, 11, 3), (6, 3, 8, 2)):
        prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
        dims = [k.dims for k in prob.stages]
        a = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, lib_path=EMU)
        b = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, lib_path=EMU)
        for t, k in enumerate(prob.stages):
            a.upload_knot(0, t, k)
        a.set_init(0, prob.G0, prob.g0)
        assert a.backward(1e-10) and a.forward()
        assert b.backward_blocks(prob, 1e-10) and b.forward()
        for A, B in zip(a.solution(0), b.solution(0)):
            for x, y in zip(A, B):
                assert np.array_equal(x, y)
        assert np.array_equal(a.download_packed(), b.download_packed())
        # the roll-out gar_hip_backward_blocks enqueues behind the sweep (no parameter) is what forward() / the bulk
        # read-back hand out: same bits as the plain sequence, also when asked twice, after collapseFeedback, and
        # after the NEXT sweep on changed data (nothing stale survives a backward)
        for x, y in zip(a.fetch_results(0), b.fetch_results(0)):
            assert np.array_equal(x, y)
        assert b.forward()
        for x, y in zip(a.fetch_results(0), b.fetch_results(0)):
            assert np.array_equal(x, y)
        a.collapse_feedback(), b.collapse_feedback()
        assert a.forward() and b.forward()
        for x, y in zip(a.fetch_results(0), b.fetch_results(0)):
            assert np.array_equal(x, y)
        before = [x.copy() for x in b.fetch_results(0)]
        for k in prob.stages:
            k.q[:] = rng.standard_normal(k.q.shape)
        for t, k in enumerate(prob.stages):
            a.upload_knot(0, t, k)
        assert a.backward(1e-10) and a.forward()
        assert b.backward_blocks(prob, 1e-10) and b.forward()
        after = b.fetch_results(0)
        assert not np.array_equal(before[0], after[0])
        for x, y in zip(a.fetch_results(0), after):
            assert np.array_equal(x, y)
    two = BatchedRiccatiSolver(dims, nx, batch=2, lib_path=EMU)
    with pytest.raises(RuntimeError, match="batch"):
        two.backward_blocks(prob, 1e-10)


def test_terminal_knot_without_successor_dimension():
    """SolverProxDDP builds its terminal knot with nx2 = 0 (solvers/proxddp/workspace.hxx:54-55); tests/gar/ builds it
    with nx2 = nx.  The specialised families address the terminal factor record through offsets that assume nx2 = nx
    rows of [yff | Aff] in it: until round 5 such a problem bound `wave<8,4>` all the same and the sweep wrote 72
    doubles past the factor records (found by running the reference's own ProxDDP loop on the backend under
    AddressSanitizer).  Now the library takes the knot in as nx2 = nx with zeros for its A, f -- which nothing of
    the algorithm reads (riccati-kernel.hxx:130-193) -- and hands its gains back with the caller's row count: same
    kernels, same answer as the nx2 = nx problem, serial, padded, in leg mode, constrained at the terminal knot."""
    from aligator_amd.gar import BatchedRiccatiSolver, lqrComputeKktError
    from aligator_amd.lqr import LqrKnot, LqrProblem

    def shrink_terminal(base):
        last = base.stages[-1]
        term = LqrKnot(last.nx, 0, last.nc, 0)        # nx2 = 0: A is 0 x nx, f empty
        for name in ("Q", "q", "C", "d"):
            getattr(term, name)[...] = getattr(last, name)
        prob = LqrProblem(base.stages[:-1] + [term], base.nc0)
        prob.G0[...] = base.G0
        prob.g0[...] = base.g0
        return prob
    cases = [(8, 4, 0, 6, 1, 1, "wave<8,4>"), (8, 4, 0, 6, 3, 1, "wave<8,4>"), (6, 3, 0, 7, 1, 1, "wave<8,4>"),
             (8, 4, 0, 9, 1, 3, "wave_leg<8,4>"), (5, 2, 2, 5, 1, 1, "generic")]
    for nx, nu, nc, N, batch, legs, want in cases:
        base = synth.generate_lq_problem(5, np.ones(nx), N, nx, nu, nc=nc, mode="W")
        prob = shrink_terminal(base)
        assert prob.stages[-1].dims == (nx, 0, nc, 0, 0)
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=batch, num_legs=legs, lib_path=EMU)
        assert s.kernel_name.startswith(want), s.kernel_name
        s.upload([prob] * batch)
        mu = 1e-8 if nc else 1e-10
        assert s.backward(mu) and s.forward()
        sol = s.solution(batch - 1)
        _, osol, ref = pc.oracle_serial(base, mu)     # (the oracle never reads the terminal A, f either)
        for a, b in zip(sol, ref):
            assert pc.maxdiff(a, b) <= 1e-8 * pc.scale_of(ref)
        assert max(lqrComputeKktError(prob, *sol, mueq=mu)) <= 1e-8 * pc.scale_of(ref)
        # the terminal knot's gains: the caller's nc rows [zff | Z], through the per-stage getter and the bulk path
        f = s.factor(N, batch - 1)
        assert f.ff.shape == (nc,) and f.fb.shape == (nc, nx)
        ffs, fbs = s.gains_all(batch - 1)
        assert ffs[N].shape == (nc,) and fbs[N].shape == (nc, nx)
        if nc:
            assert pc.maxdiff([f.ff], [osol.datas(N).ff[:nc]]) <= 1e-9 * max(1.0, float(np.abs(f.ff).max()))
            assert np.array_equal(f.ff, ffs[N]) and np.array_equal(f.fb, fbs[N])
        s.close()


def test_terminal_knot_without_successor_dimension_through_the_solver_classes(monkeypatch):
    """the same problem shape through the host mirror's solver classes (gar_hip_upload_stage knot by knot, the
    per-stage getters) and with the legs over two (virtual) devices behind one handle"""
    from aligator_amd.gar import (ParallelRiccatiSolver, ProximalRiccatiSolver, lqrComputeKktError,
                                  lqrInitializeSolution)
    from aligator_amd.lqr import LqrKnot, LqrProblem
    monkeypatch.setenv("GAR_EMU_DEVICES", "2")
    nx, nu, N = 8, 4, 9
    base = synth.generate_lq_problem(11, np.ones(nx), N, nx, nu, mode="W")
    term = LqrKnot(nx, 0, 0, 0)
    term.Q[...] = base.stages[-1].Q
    term.q[...] = base.stages[-1].q
    prob = LqrProblem(base.stages[:-1] + [term], nx)
    prob.G0[...] = base.G0
    prob.g0[...] = base.g0
    _, _, ref = pc.oracle_serial(base, 1e-10)
    ser = ProximalRiccatiSolver(prob, lib_path=EMU)
    assert ser.kernel_name == "wave<8,4>"
    ser.backward(1e-10)
    sol = lqrInitializeSolution(prob)
    ser.forward(*sol)
    for a, b in zip(sol, ref):
        assert pc.maxdiff(a, b) <= 1e-10 * pc.scale_of(ref)
    assert ser.datas[N].ff.shape == (0,) and ser.datas[N].fb.shape == (0, nx)
    assert ser.getFeedback(0).shape[1] == nx and max(lqrComputeKktError(prob, *sol, mueq=1e-10)) <= 1e-9
    import ctypes as C
    L = C.CDLL(EMU)
    L.emu_set_device_count(2)
    try:
        par = ParallelRiccatiSolver(prob.copy(), 3, lib_path=EMU, devices=[0, 1])
        assert par.kernel_name.startswith("wave_leg<8,4>")
        par.backward(1e-10)
        psol = lqrInitializeSolution(prob)
        par.forward(*psol)
        for a, b in zip(psol, ref):
            assert pc.maxdiff(a, b) <= 1e-9 * pc.scale_of(ref)
    finally:
        L.emu_set_device_count(1)


def test_behaviour_switches_through_the_api():
    """gar_hip_set_option: what the GAR_HIP_* environment variables choose can be chosen through the ABI (the
    reference's knobs are struct fields, parallel-solver.hpp:92-94); the call takes precedence over the environment,
    None hands the name back to it, an unknown name is refused"""
    from aligator_amd.gar import BatchedRiccatiSolver, set_option, get_option
    nx, nu, N = 8, 4, 5
    prob = synth.generate_lq_problem(3, np.ones(nx), N, nx, nu, mode="W")
    dims = [k.dims for k in prob.stages]

    def kernel():
        s = BatchedRiccatiSolver(dims, nx, batch=1, lib_path=EMU)
        name = s.kernel_name
        s.upload([prob])
        assert s.backward(1e-10) and s.forward()
        sol = s.solution(0)
        s.close()
        return name, sol
    base, sol0 = kernel()
    assert base == "wave<8,4>" and get_option("BACKWARD", EMU) is None
    try:
        set_option("BACKWARD", "wg4", EMU)
        assert get_option("GAR_HIP_BACKWARD", EMU) == "wg4"
        n1, sol1 = kernel()
        assert n1 == "mfma<8,4>"
        set_option("GAR_HIP_FORCE_GENERIC", "1", EMU)
        n2, sol2 = kernel()
        assert n2 == "generic"
        for a, b, c in zip(sol0, sol1, sol2):
            assert pc.maxdiff(a, b) <= 1e-12 and pc.maxdiff(a, c) <= 1e-10
        with pytest.raises(ValueError, match="unknown switch"):
            set_option("NO_SUCH_SWITCH", "1", EMU)
    finally:
        set_option("BACKWARD", None, EMU)
        set_option("FORCE_GENERIC", None, EMU)
    assert kernel()[0] == base

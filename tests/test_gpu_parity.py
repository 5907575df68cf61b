"""Parity of the HIP path (through the C ABI, on a real MI355X) against the CPU
oracle, at the reference's own test sizes, plus size-independent properties at
BASELINE.json's full sizes.  Run with ``pytest -m gpu``."""
import os
import numpy as np
import pytest

from aligator_amd import synth
import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("horz", [4, 8, 16])
def test_riccati_short_horz_pb(horz):                         # tests/gar/riccati.cpp:26-85
    pc.check_serial(synth.short_horizon_problem(horz), 1e-14, 1e-9, kkt_tol=1e-9)


def test_riccati_one_knot_prob():                             # riccati.cpp:87-105
    prob = synth.generate_lq_problem(1, np.zeros(2), 0, 2, 2)
    pc.check_serial(prob, 1e-13, 1e-10, kkt_tol=1e-10)


@pytest.mark.parametrize("horz", [20, 100])
@pytest.mark.parametrize("mode", ["F", "W"])
def test_riccati_random_large_problem(horz, mode):            # riccati.cpp:107-139
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(42, np.zeros(nx), horz, nx, nu, mode=mode)
    pc.check_serial(prob, 1e-14, pc.TOL[mode], kkt_tol=1e-6 if mode == "F" else 1e-9)


def test_riccati_parametric():                                # riccati.cpp:157-192
    rng = np.random.default_rng(9)
    nx, nu, nth = 10, 4, 1
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 100, nx, nu, nth=nth)
    theta = rng.uniform(-1, 1, nth)
    solver, _, _ = pc.check_serial(prob, 1e-12, 1e-9, theta=theta, kkt_tol=1e-9)
    for arr in (solver.kkt0.ff, solver.kkt0.fth, solver.thGrad, solver.thHess,
                solver.datas[0].vm.vt, solver.datas[100].vm.Vtt):
        assert np.isfinite(arr).all()


def test_constrained_bench_shape():
    """bench/gar-riccati.cpp:19-22: nc=32, mu=1e-11 (multipliers O(1/mu))."""
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 16, nx, nu, nc=nc, mode="W")
    solver, _, _ = pc.check_serial(prob, 1e-11, 1e-7, factors=False)
    assert solver.kernel_name == "wave<36,12,32>"     # the constrained wave kernels, not the generic ones


@pytest.mark.parametrize("legs", [2, 3, 4, 6])
def test_parallel_solver_on_the_reference_bench_shape_nc32(legs):
    """bench/gar-riccati.cpp:64-90 (BM_parallel<2,3,4,6>): nx = 36, nu = 12, nc = 32, leg mode."""
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 32, nx, nu, nc=nc, mode="W")
    par = pc.check_parallel(prob, 1e-8, legs, 1e-7)
    assert par._impl.kernel_name.startswith("wave_leg<36,12>+fold")      # round 3: no longer on the generic leg kernels


@pytest.mark.parametrize("legs", [2, 6])
def test_parallel_solver_on_the_reference_bench_shape_nc32_at_benchmark_size(legs):
    """BM_parallel at the benchmark's own size and mu (bench/gar-riccati.cpp:19-22, 64-90: N = 256, mu = 1e-11): the
    multipliers are O(1/mu), so v and lambda are judged against what the problem's conditioning allows
    (pc.check_parallel(conditioned=True): the oracle's own serial / leg-parallel / LAPACK spread), x and u at 1e-8."""
    nx, nu, nc = 36, 12, 32
    rng = np.random.default_rng(19)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 256, nx, nu, nc=nc, mode="W")
    rep = {}
    par = pc.check_parallel(prob, 1e-11, legs, 1e-8, conditioned=True, report=rep)
    assert par._impl.kernel_name.startswith("wave_leg<36,12>+fold")
    assert max(rep["hip_leg-oracle_leg"][:2]) <= 1e-10
    print(f"nc=32 N=256 mu=1e-11 legs={legs}: hip-oracle_leg {rep['hip_leg-oracle_leg']} tolerances {rep['tolerances']}")


def test_constrained_legs_fold_onto_the_wave_leg_kernels():
    pc.check_constrained_legs_fold(shapes=((8, 4, 4, 41, 5, 1e-6), (16, 8, 8, 30, 4, 1e-7), (36, 12, 32, 24, 3, 1e-7)))


def test_coupled_constraints_in_leg_mode_on_the_constrained_segment_legs():
    """Round 6 (gar_cstr_seg.hpp): D != 0 in leg mode on the serial constrained chain's stage kernels, leg by leg, + the
    parameter recursion -- incl. the reference's benchmark shape (36, 12, 32), N = 64 / 8 legs."""
    pc.check_constrained_legs_segments(shapes=((8, 4, 4, 41, 5, 1e-6), (16, 8, 8, 30, 4, 1e-7), (36, 12, 32, 64, 8, 1e-7)))


SOAK_FAILURES = [(101, 353783436, None, 42, 2), (2026, 755480262, "constrained", 22, 8)]


@pytest.mark.parametrize("soak_seed,inner_seed,focus,horz,legs", SOAK_FAILURES)
def test_soak_failures_replayed_and_arbitrated(soak_seed, inner_seed, focus, horz, legs, capsys):
    """The two round-2 soak failures (profiles/r02_soak_gpu.log:8, r02_soak_constrained_gpu.log:2; nx = 36,
    nu = 12, nc = 32 in leg mode at mu = 1e-8), replayed exactly (tests/soak_draws.py) and arbitrated four
    ways: HIP-leg, oracle-leg, oracle-serial, LAPACK on the global dense KKT matrix.  The log line says which
    pairs disagree.  Finding (same on the wave emulator): x and u agree to 1e-15 between ALL FOUR; v and lambda
    (of order 1/mu) differ by 1e-7 ... 2e-6 relative between ANY two of them, oracle-serial vs LAPACK
    included -- conditioning, not a kernel: HIP-leg is no outlier.  The check therefore bounds HIP-leg by what
    the problem's conditioning allows (pc.check_parallel(conditioned=True)) -- and x, u at the plain tolerance."""
    from soak_draws import find_draw
    d = find_draw(soak_seed, inner_seed, focus)
    assert (d["nx"], d["nu"], d["nc"], d["horz"], d["legs"]) == (36, 12, 32, horz, legs)
    mu = max(d["mu"], 1e-8)                                     # the value the failing run used
    rep = {}
    pc.check_parallel(d["prob"], mu, legs, 1e-8, conditioned=True, report=rep)
    with capsys.disabled():
        print(f"\nsoak replay seed={inner_seed} N={horz} legs={legs} mu={mu:.1e} scale={rep['scale']:.2e}  (x, u, v, lambda) relative:")
        for k in ("hip_leg-oracle_leg", "hip_leg-oracle_serial", "hip_leg-lapack", "oracle_leg-oracle_serial",
                  "oracle_leg-lapack", "oracle_serial-lapack"):
            print(f"  {k:26s}", " ".join(f"{v:.2e}" for v in rep[k]))
        print(f"  kkt/scale: hip_leg {rep['kkt_hip_leg']:.2e} oracle_leg {rep['kkt_oracle_leg']:.2e}")
    # x and u are well determined: every pair within 1e-12
    for k in ("hip_leg-oracle_leg", "hip_leg-oracle_serial", "hip_leg-lapack"):
        assert max(rep[k][:2]) <= 1e-12, (k, rep[k])
    # HIP-leg is no outlier: no farther from LAPACK than 4 x the worst of the CPU solvers among themselves
    cpu = max(max(rep[k][2:]) for k in ("oracle_leg-oracle_serial", "oracle_leg-lapack", "oracle_serial-lapack"))
    assert max(rep["hip_leg-lapack"][2:]) <= 4 * cpu, (rep["hip_leg-lapack"], cpu)


SOAK_R3 = [(301, 234238820, None), (301, 917161254, None), (302, 331723640, None), (303, 703800132, "constrained")]


@pytest.mark.parametrize("soak_seed,inner_seed,focus", SOAK_R3)
def test_round3_soak_failures_replayed(soak_seed, inner_seed, focus):
    """The four draws of the round-3 soak (profiles/r03_soak_gpu.log; 9 015 problems) that missed the FIRST version
    of the conditioning-aware tolerance -- all constrained with D != 0 and mu <= 2e-9: (6,3,2) on the generic
    kernels (two in leg mode failing a stage-FACTOR comparison at 1e-6 while the solution agreed to 1e-7; one serial
    at hip-oracle 8e-7 where oracle-LAPACK is 1.0e-6), (16,8,8) in leg mode at the edge of 4 x the CPU spread.
    Replayed exactly (tests/soak_draws.py, version 2) under the policy the soak now applies."""
    from soak_draws import draws
    for i, d in enumerate(draws(soak_seed, focus, version=2)):
        if d["seed"] == inner_seed:
            break
        assert i < 20000
    if d["legs"] == 1:
        pc.check_serial(d["prob"], d["mu"], 1e-6, factors=False, conditioned=True)
    else:
        pc.check_parallel(d["prob"], d["mu"], d["legs"], 1e-6, conditioned=True)


def test_round4_soak_failure_replayed():
    """The one failing draw of the round-4 soak (2 518 + 424 problems; SOAK_SEED 2026, inner seed 423562352): (8, 4, 4),
    N = 36, 8 legs, mu = 1.6e-9, constraints folded onto the wave-leg kernels.  Arbitrated (scripts/soak_arbitrate.py):
    block cyclic reduction of the condensed system left a backward error of 9.4e-13 -- under the 1e-12 gate of rounds
    3 -- and with it multipliers 150 x farther from LAPACK than any CPU solver (x, u: 2e-11); the round-3 library does
    the same.  The gate is 1e-13 now: the solve is handed to the chain in the reference's order (omega 4e-17)."""
    from soak_draws import draws
    for i, d in enumerate(draws("2026", None, version=2)):
        if d["seed"] == 423562352:
            break
        assert i < 20000
    os.environ["GAR_HIP_BACKWARD"], os.environ["GAR_HIP_WIDE"] = d["backward"], d["wide"]
    try:
        rep = {}
        par = pc.check_parallel(d["prob"], d["mu"], d["legs"], 1e-8, conditioned=True, report=rep)
        assert par._impl.kernel_name.startswith("wave_leg<8,4>+fold") and par._impl.condensed_resolved(0)
        assert max(rep["hip_leg-lapack"][:2]) <= 1e-12 and max(rep["hip_leg-lapack"][2:]) <= 1e-5
    finally:
        os.environ.pop("GAR_HIP_BACKWARD", None)
        os.environ.pop("GAR_HIP_WIDE", None)


def test_constrained_bench_shape_full_factors_at_benchmark_size():
    """The reference's own benchmark configuration at its benchmark SIZE and mu (bench/gar-riccati.cpp:19-22,
    53-62: nx = 36, nu = 12, nc = 32, N = 256, mu = 1e-11): every factor block of every stage, kkt0 and the
    solution against the oracle.  Multipliers and Vxx are O(1/mu) here; errors are relative to each block's
    scale (the reference's bench never checks residuals)."""
    nx, nu, nc, N, mu = 36, 12, 32, 256, 1e-11
    rng = np.random.default_rng(19)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), N, nx, nu, nc=nc, mode="W")
    solver, sol, ref = pc.check_serial(prob, mu, 1e-7, factors=True)
    assert solver.kernel_name == "wave<36,12,32>"
    bound, _ = pc.conditioning_bound(prob, mu, ref)
    sc = pc.scale_of(ref)
    err = [pc.maxdiff(a, b) / sc for a, b in zip(sol, ref)]
    print(f"nc=32 N=256 mu=1e-11: hip-oracle {err}, oracle-lapack {bound}, scale {sc:.2e}")


def test_constrained_decoupled_dense_c_and_alternating_d():
    """D = 0 stages (gar_wave2.hpp, NC > 0) with a dense C, and sweeps that alternate between the decoupled
    stage and the (NU+NC) Bunch-Kaufman stage."""
    pc.check_constrained_decoupled(shapes=((8, 4, 4, 25, 1e-6), (16, 8, 8, 30, 1e-8), (36, 12, 32, 40, 1e-8)))


@pytest.mark.parametrize("nx,nu,nc,horz,mu", [(36, 12, 32, 40, 1e-6), (16, 8, 8, 30, 1e-8), (8, 4, 4, 25, 1e-5)])
def test_constrained_wave_kernels_factors(monkeypatch, nx, nu, nc, horz, mu):
    """Constrained stages on the wave kernels: every factor block (ff = [kff; zff; yff],
    fb = [K; Z; Aff], Vxx, vx), kkt0 and the solution against the oracle -- and against the generic
    kernels on the same problem."""
    from aligator_amd.gar import ProximalRiccatiSolver, lqrInitializeSolution
    rng = np.random.default_rng(600 + nx)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nc=nc, mode="W")
    solver, sol, _ = pc.check_serial(prob, mu, 1e-8)
    assert solver.kernel_name == f"wave<{nx},{nu},{nc}>"
    monkeypatch.setenv("GAR_HIP_FORCE_GENERIC", "1")
    g = ProximalRiccatiSolver(prob)
    assert g.kernel_name == "generic"
    g.backward(mu)
    sg = lqrInitializeSolution(prob)
    g.forward(*sg)
    sc = pc.scale_of(sg)
    for A, B in zip(sol, sg):
        assert pc.maxdiff(A, B) <= 1e-9 * sc


def test_constrained_small_with_2x2_pivots():
    rng = np.random.default_rng(5)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(6), 5, 6, 3, nc=4, mode="W")
    pc.check_serial(prob, 1e-9, 1e-9)


def test_north_star_config_single_problem():
    """BASELINE.json configs[1]: N=256, nx=36, nu=12, fp64, serial in time."""
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(7, np.zeros(nx), 256, nx, nu, mode="W")
    pc.check_serial(prob, 1e-14, 1e-9, kkt_tol=1e-9)


@pytest.mark.parametrize("nthreads", [2, 6])
def test_parallel_solver_class(nthreads):                     # tests/gar/parallel.cpp:185-245
    rng = np.random.default_rng(17)
    nx, nu, horizon = 32, 12, 50
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horizon, nx, nu)
    pc.check_parallel(prob, 1e-9, nthreads, 1e-7, rounds=3, rng=rng)


def test_parallel_config3():
    """BASELINE.json configs[2]: N=1024, nx=12, nu=6, legs on one GPU."""
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, np.zeros(12), 1024, 12, 6, mode="W")
    for legs in (8, 64):
        pc.check_parallel(prob, 1e-9, legs, 1e-7)


def test_batched_problems_and_legs():
    probs = [synth.generate_lq_problem(100 + i, np.zeros(36), 32, 36, 12, mode="W") for i in range(5)]
    pc.check_batched(probs, 1e-12, 1e-9)
    pc.check_batched(probs, 1e-12, 1e-8, num_legs=4)


def test_failed_factorisation_raises():
    from aligator_amd.gar import ProximalRiccatiSolver
    prob = synth.generate_lq_problem(3, np.zeros(3), 3, 3, 2, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    with pytest.raises(RuntimeError, match="LDL"):
        ProximalRiccatiSolver(prob).backward(1e-10)


def test_full_size_batch_properties():
    """Size-independent properties at the bench's full sizes (device-generated
    batch): KKT residual of sampled problems, idempotence of a repeated sweep,
    and agreement with the oracle on the sampled problems."""
    import torch
    from aligator_amd import synth_device
    from aligator_amd.gar import BatchedRiccatiSolver, lqrComputeKktError
    nx, nu, N, B = 36, 12, 256, 64
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=B)
    synth_device.fill_problems(s, seed=99, mode="W", keep=(0, 31, -1))
    s.backward(1e-14)
    s.forward()
    first = [s.solution(b) for b in (0, 31, B - 1)]
    s.backward(1e-14)
    s.forward()
    for (b, sol) in zip((0, 31, B - 1), first):
        again = s.solution(b)
        for A, C in zip(sol, again):
            assert pc.maxdiff(A, C) == 0.0                     # deterministic / idempotent
        prob = synth_device.download_problem(s, b)
        assert max(lqrComputeKktError(prob, *sol, mueq=1e-14)) <= 1e-9
        _, _, ref = pc.oracle_serial(prob, 1e-14)
        for A, C in zip(sol, ref):
            assert pc.maxdiff(A, C) <= 1e-9 * pc.scale_of(ref)
    assert s.num_failed() == 0
    del torch


class _BatchDatas:
    """datas[t] of problem b of a BatchedRiccatiSolver (what compare_factors indexes)."""

    def __init__(self, solver, b):
        self.s, self.b = solver, b

    def __getitem__(self, t):
        return self.s.factor(t, self.b)


@pytest.mark.parametrize("mode,horz", [("W", 256), ("F", 100), ("F", 256)])
def test_headline_wave_kernel_full_parity(mode, horz):
    """The kernel the bench line is measured on -- gar_backward_wave<36,12>, which the library selects
    once batch > #CUs -- at the north-star size (N=256; the reference's own generator F also at its
    test size N=100, tests/gar/riccati.cpp:107-139, and bar 1e-6): eight problems spread over a
    1024-problem device-generated batch, EVERY stage's ff / fb / Vxx / vx, kkt0 and the whole
    solution against the oracle; plus the slow-path counters the bench line reports."""
    from aligator_amd import synth_device
    from aligator_amd.gar import BatchedRiccatiSolver, lqrComputeKktError
    nx, nu, B, mueq = 36, 12, 1024, 1e-14
    dims = [(nx, nu, 0, nx, 0)] * horz + [(nx, 0, 0, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=B)
    assert s.kernel_name == "wave<36,12>"
    keep = tuple(int(b) for b in np.linspace(0, B - 1, 8))
    synth_device.fill_problems(s, seed=2024 + horz, mode=mode, keep=keep)
    assert s.backward(mueq) and s.forward()
    slow, pivoted = s.slow_path_stages()
    assert 0 <= pivoted <= slow <= B * horz
    tol = pc.TOL[mode]
    for b in keep:
        prob = synth_device.download_problem(s, b)
        _, osol, ref = pc.oracle_serial(prob, mueq)
        sol = s.solution(b)
        sc = pc.scale_of(ref)
        for A, C in zip(sol, ref):
            assert pc.maxdiff(A, C) <= tol * sc, b
        assert max(lqrComputeKktError(prob, *sol, mueq=mueq)) <= (1e-9 if mode == "W" else 1e-6) * sc
        pc.compare_factors(_BatchDatas(s, b), osol, horz, tol, names=("ff", "fb"), vnames=("Vxx", "vx"))
        ff0, _, _, _ = s.initial(b)
        assert np.abs(ff0 - osol.kkt0_ff).max() <= tol * max(1.0, np.abs(osol.kkt0_ff).max())
    print(f"wave<36,12> {mode} N={horz}: slow-path stages {slow} / {B * horz}, pivoted {pivoted}")


def test_config4_shape_legs_on_one_gpu():
    """BASELINE.json configs[3] shape (N=2048, nx=36, nu=12) with the 8-way leg partition
    the 8-GPU run uses, all legs on this GPU: same condensed system, same forward."""
    prob = synth.generate_lq_problem(31, np.zeros(36), 2048, 36, 12, mode="W")
    pc.check_parallel(prob, 1e-9, 8, 1e-7)


@pytest.mark.parametrize("nx,nu,horz,legs", [(36, 12, 256, 32), (36, 12, 257, 7), (32, 12, 64, 8),
                                              (16, 8, 100, 16), (12, 4, 60, 5), (8, 4, 40, 12)])
def test_wave_leg_kernels_and_cyclic_reduction(nx, nu, horz, legs):
    """Parallel-in-time at kernel speed (csrc/gar_wave_leg.hpp, csrc/gar_cyclic.hpp): per-stage
    factors against the oracle's own leg-parallel solver, solution against the serial oracle,
    collapseFeedback, a re-solve after the problem changed (tests/gar/parallel.cpp:185-245)."""
    rng = np.random.default_rng(4000 + nx + legs)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    par = pc.check_parallel(prob, 1e-10, legs, 1e-9, rounds=1, rng=rng)
    assert par._impl.kernel_name == f"wave_leg<{nx},{nu}>"


def test_wave_leg_kernels_reference_faithful_generator():
    """The reference's own random generator (tests/gar/test_util.cpp: dense random A, B -- value
    functions grow along the horizon) at its own bar for nx = 36 (tests/gar/riccati.cpp:138)."""
    rng = np.random.default_rng(4300)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(36), 40, 36, 12, mode="F")
    par = pc.check_parallel(prob, 1e-9, 5, pc.TOL["F"], rounds=1, rng=rng)
    assert par._impl.kernel_name == "wave_leg<36,12>"


def test_condensed_solvers_agree_on_gpu(monkeypatch):
    """Block cyclic reduction (no fallback: refinement off), the elimination chain and the generic
    kernel on the north-star shape, 64 legs, a batch of problems; and the gated fallback."""
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N, legs = 36, 12, 512, 64
    probs = [synth.generate_lq_problem(4100 + i, np.full(nx, 0.1 * i), N, nx, nu, mode="W") for i in range(3)]
    dims = [k.dims for k in probs[0].stages]

    def solve(refine=None, thr=1e-10, backward_ok=None):
        s = BatchedRiccatiSolver(dims, nx, batch=len(probs), num_legs=legs)
        if refine is not None:
            s.set_refinement(thr, refine, backward_ok)
        s.upload(probs)
        assert s.backward(1e-10) and s.forward()
        return s, [s.solution(b) for b in range(len(probs))]

    s, cyc = solve(refine=0)
    assert s.kernel_name == "wave_leg<36,12>"
    for b in range(len(probs)):
        resid, steps = s.condensed_info(b)
        assert resid < 1e-9 and steps == 0
    s2, gated = solve(refine=2, thr=1e-300, backward_ok=0.0)
    assert s2.condensed_info(0)[1] == 2          # the chain kernel (with refinement) took over
    s3, _ = solve(refine=2, thr=1e-300)          # default gate: the cyclic-reduction solve stands on its backward error
    assert s3.condensed_info(0)[1] == 0 and 0.0 < s3.condensed_backward_error(0) <= 1e-13
    monkeypatch.setenv("GAR_HIP_CONDENSED", "chain")
    _, chain = solve()
    monkeypatch.setenv("GAR_HIP_CONDENSED", "generic")
    _, gen = solve()
    for b, prob in enumerate(probs):
        _, _, ref = pc.oracle_serial(prob, 1e-10)
        sc = pc.scale_of(ref)
        for sol in (cyc[b], gated[b], chain[b], gen[b]):
            for A, B in zip(sol, ref):
                assert pc.maxdiff(A, B) <= 1e-9 * sc


def test_condensed_block_inverse_bunch_kaufman_fallback():
    pc.check_condensed_block_inverse_fallback()


@pytest.mark.parametrize("leg_waves", ["2", "1"])
def test_leg_kernels_bunch_kaufman_fallback_and_failure(monkeypatch, leg_waves):
    monkeypatch.setenv("GAR_HIP_LEG_WAVES", leg_waves)
    pc.check_leg_kernels_bunch_kaufman_fallback()
    # and the one-wave-per-leg kernel against the two-wave default on the north-star shape
    prob = synth.generate_lq_problem(4400, np.ones(36), 64, 36, 12, mode="W")
    par = pc.check_parallel(prob, 1e-10, 8, 1e-9)
    assert par._impl.kernel_name == "wave_leg<36,12>"


def test_sharded_solver_single_rank_rccl():
    """aligator_amd.sharded on the real device path: torch views of the library's device
    buffers, all_gather_into_tensor over RCCL (world_size 1 is all a 1-GPU box offers; the
    2-rank flow is covered on CPU by tests/test_sharded_gloo.py)."""
    import torch
    import torch.distributed as dist
    from aligator_amd.sharded import ShardedRiccatiSolver
    from aligator_amd.gar import lqrComputeKktError
    torch.cuda.set_device(0)
    # (a one-rank group needs no rendezvous socket: an in-process store, so no port can be in use)
    dist.init_process_group("nccl", store=dist.HashStore(), world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        for (nx, nu) in ((12, 6), (36, 12)):  # generic leg kernels ; wave-leg kernels + cyclic reduction
            prob = synth.generate_lq_problem(32, np.zeros(nx), 255, nx, nu, mode="W")
            _, _, ref = pc.oracle_serial(prob, 1e-9)
            for legs in (2, 8):
                s = ShardedRiccatiSolver([k.dims for k in prob.stages], prob.nc0, legs, batch=1)
                s.impl.upload([prob])
                s.backward(1e-9)
                s.forward()
                sol = s.gather_solution(0)
                sc = pc.scale_of(ref)
                for A, B in zip(sol, ref):
                    assert pc.maxdiff(A, B) <= 1e-7 * sc
                assert max(lqrComputeKktError(prob, *sol, mueq=1e-9)) <= 1e-7 * sc
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("family,name", [("wave", "wave"), ("wg4", "mfma")])
@pytest.mark.parametrize("nx,nu,horz,mode", [(36, 12, 33, "W"), (36, 12, 20, "F"), (32, 12, 17, "W"),
                                             (16, 8, 12, "F"), (12, 4, 9, "W"), (8, 4, 5, "W")])
def test_both_backward_families(monkeypatch, family, name, nx, nu, horz, mode):
    """The throughput kernel (one wave per problem) and the latency kernel (one workgroup per
    problem) are selected by batch size; force each and check every specialised shape."""
    monkeypatch.setenv("GAR_HIP_BACKWARD", family)
    prob = synth.generate_lq_problem(700 + nx + horz, np.zeros(nx), horz, nx, nu, mode=mode)
    solver, _, _ = pc.check_serial(prob, 1e-12, pc.TOL[mode], kkt_tol=1e-6 if mode == "F" else 1e-9)
    assert solver.kernel_name == f"{name}<{nx},{nu}>"


def test_wave_family_bunch_kaufman_fallback_and_failure(monkeypatch):
    """Stages where Bunch-Kaufman pivots (out-of-line exact path) and a zero pivot column
    (the reference throws, riccati-kernel.hxx:239-241) on the throughput kernel."""
    from aligator_amd.gar import ProximalRiccatiSolver
    monkeypatch.setenv("GAR_HIP_BACKWARD", "wave")
    nx, nu = 8, 4
    prob = synth.generate_lq_problem(11, np.zeros(nx), 4, nx, nu, mode="W")
    for k in prob.stages[:-1]:
        k.R[...] = np.array([[1e-3, 2.0, 0.1, 0.0], [2.0, 1e-3, 0.0, 0.1],
                             [0.1, 0.0, 3.0, 0.2], [0.0, 0.1, 0.2, 4.0]])
        k.B[...] *= 1e-2
    solver, _, _ = pc.check_serial(prob, 1e-12, 1e-9)
    assert solver.kernel_name == "wave<8,4>"
    bad = synth.generate_lq_problem(3, np.zeros(8), 3, 8, 4, mode="W")
    for k in bad.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    with pytest.raises(RuntimeError, match="LDL"):
        ProximalRiccatiSolver(bad).backward(1e-10)


@pytest.mark.parametrize("nx,nu,horz,legs", [(36, 12, 64, 1), (36, 12, 64, 8), (12, 6, 40, 1), (30, 10, 33, 1)])
def test_bulk_gains_and_solution_readback(nx, nu, horz, legs):
    """gar_hip_fetch_results / gar_hip_get_gains_all: one gather + one copy, bitwise what the per-stage
    calls return (wave / wave-leg kernels, generic kernels, a padded shape)."""
    prob = synth.generate_lq_problem(300 + nx, np.ones(nx), horz, nx, nu, mode="W")
    pc.check_bulk_gains(prob, 1e-10, num_legs=legs)


def test_wave_kernel_second_bunch_kaufman_test():
    pc.check_second_bunch_kaufman_test()


def test_constrained_wave_kernels_bunch_kaufman_pivoting():
    pc.check_constrained_pivoting(shapes=((8, 4, 4, 6, 1e-6), (16, 8, 8, 5, 1e-7), (36, 12, 32, 12, 1e-6)))


def _bench_line(cmd):
    """cmd: the argument list, or a function of a port probed free on every local address (launches through
    torch.distributed.run: a rendezvous that finds the port taken after all is tried again on another one)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for attempt in range(3):
        argv = cmd
        if callable(cmd):
            with socket.socket() as s:
                s.bind(("", 0))
                argv = cmd(s.getsockname()[1])
        r = subprocess.run([sys.executable] + argv, cwd=root, capture_output=True, text=True, timeout=900)
        if r.returncode == 0 or not callable(cmd) or "address already in use" not in (r.stdout + r.stderr):
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line on rank 0
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    """bench.py's JSON line: the driver's contract fields, the roofline and cpu_baseline objects."""
    d = _bench_line(["bench.py", "--batch", "128", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "sweeps/s"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert "workload" in d["config"] and d["value"] > 0
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["parity"]["max_rel_err_vs_oracle"] < 1e-9 and d["parity"]["failed_factorisations"] == 0
    assert d["parallel_in_time"]["max_rel_diff_vs_serial"] < 1e-9
    sc = d["roofline"]["stream_ceiling"]   # the sweep's bytes alone, same run: the kernel cannot beat it by much
    assert sc["ms"] > 0 and 0.3 < sc["frac_of_peak"] < 1.0 and sc["kernel_over_stream"] > 0.5
    sec = d["secondary_shapes"]   # the reference's own benchmark shape and the Talos-walk LQ shape
    assert sec["reference_bench_shape_nc32"]["kernel"] == "wave<36,12,32>" and sec["talos_walk_lq_shape"]["kernel"] == "pair<56,24>"
    for v in sec.values():
        assert v["sweeps_per_s"] > 0 and v["failed_factorisations"] == 0 and 0 < v["backward_frac_of_hbm_roofline"] < 1
        # parity of what was timed, in the same run (relative to the largest multiplier: O(1/mu) at nc = 32)
        assert v["max_rel_err_vs_oracle"] < 1e-7 and v["max_kkt_rel"] < 1e-9
    assert sec["talos_walk_lq_shape"]["max_rel_err_vs_oracle"] < 1e-9
    cp = sc["plain_copy"]            # the plain 16 B/lane copy of the same bytes, same process
    assert cp["ms"] > 0 and 0.3 < cp["frac_of_peak"] < 1.0
    # roofline.traffic: collected in this run when rocprofv3 is there (a dict describes how), else the committed file
    assert isinstance(d["roofline"]["traffic_source"], (dict, str))


def test_bench_two_ranks_on_one_gpu():
    """The N > 1 launch line of the driver (torch.distributed.run, one rank per GPU) with two ranks
    sharing this box's only GPU (gloo barrier): whole-job value = all ranks' sweeps / max time."""
    d = _bench_line(lambda port: ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                                  "--steps", "2", "--warmup", "1", "--batch", "128", "--backend", "gloo", "--same-device"])
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d and "parallel_in_time" not in d
    assert abs(d["value"] - 2 * 128 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun around it (WORLD_SIZE unset): bench.py launches its own two ranks
    (torch.distributed.run on 127.0.0.1) and reports n_gpus = 2 with the horizon-sharded figure (two ranks share
    this box's GPU, gloo + host-staged exchange).  Asking for more GPUs than are visible fails loudly."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128",
                        "--backend", "gloo", "--same-device", "--single-generator"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    hs = d["horizon_sharded"]
    assert hs["ranks"] == 2 and hs["legs_per_rank"] == 128 and hs["max_rel_diff_vs_serial_on_rank0_stages"] < 1e-9
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n + 1), "--steps", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in (r.stdout + r.stderr)


@pytest.mark.parametrize("nx,nu,horz,family,dense", [(36, 12, 24, "wave", False), (36, 12, 24, "wg4", False),
                                                     (16, 8, 9, "wave", False), (10, 3, 12, None, False),
                                                     (12, 6, 10, None, True)])
def test_cycle_append_is_a_ring(nx, nu, horz, family, dense):
    """cycleAppend on ProximalRiccatiSolver's kernels (both specialised families, a padded shape, the
    stage-dense solver): no record moves, only the new knot is uploaded, the ring wraps."""
    pc.check_cycle_append_ring(nx=nx, nu=nu, horz=horz, cycles=horz + 3, family=family, dense=dense)


def test_bench_horizon_mode_one_rank_rccl():
    """bench.py --mode horizon (BASELINE.json configs[3]: one N=2048 problem, horizon sharded over the
    ranks, ONE RCCL all-gather per sweep, no host synchronisation inside a sweep) with the 1-rank group
    a 1-GPU box offers; the 2-rank flow runs on CPU (tests/test_sharded_gloo.py) and below when the
    box has two GPUs."""
    d = _bench_line(["bench.py", "--mode", "horizon"])
    hs = d["horizon_sharded"]
    assert d["n_gpus"] == 1 and hs["ranks"] == 1 and hs["legs"] == 256 and hs["host_syncs_per_sweep"] == 0
    assert hs["kernel"] == "wave_leg<36,12>" and hs["max_rel_diff_vs_serial_on_rank0_stages"] < 1e-9
    assert 0.0 < hs["ms_per_sweep"] < 5.0 and hs["all_gather_ms"] < hs["ms_per_sweep"]
    assert d["horizon_sharded_N16384"]["max_rel_diff_vs_serial_on_rank0_stages"] < 1e-8


def test_bench_horizon_mode_two_ranks_rccl():
    """The same with a real 2-rank RCCL all-gather; skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _bench_line(lambda port: ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                                  "--mode", "horizon"])
    hs = d["horizon_sharded"]
    assert d["n_gpus"] == 2 and hs["legs_per_rank"] == 128 and hs["max_rel_diff_vs_serial_on_rank0_stages"] < 1e-9


def _lq_from_blocks(A, B, c, Q, R, q, r, Qt, x0, horz):
    """The LQ problem ProxDDP hands to gar for a linear-quadratic trajectory problem with x0 given."""
    from aligator_amd.lqr import LqrKnot, LqrProblem
    nx, nu = B.shape
    knots = []
    for _ in range(horz):
        k = LqrKnot(nx, nu)
        k.Q[...], k.R[...], k.q[...], k.r[...] = Q, R, q, r
        k.A[...], k.B[...], k.f[...] = A, B, c
        knots.append(k)
    kt = LqrKnot(nx, 0)
    kt.Q[...] = Qt
    knots.append(kt)
    prob = LqrProblem(knots, nx)
    prob.G0[...] = -np.eye(nx)
    prob.g0[...] = x0
    return prob


@pytest.mark.parametrize("nx,nu,horz,legs,kernel", [(30, 10, 40, 4, "32,12"), (10, 3, 33, 3, "12,4"),
                                                    (13, 5, 25, 2, "16,8"), (7, 2, 19, 4, "8,4")])
def test_padded_states_and_controls(nx, nu, horz, legs, kernel):
    """Uniform unconstrained shapes without a kernel of their own: padded by the host mirror onto the
    smallest specialised shape (dummy states pinned through G0, dummy controls), results stripped."""
    rng = np.random.default_rng(nx * 7 + nu)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, mode="W")
    s, _, _ = pc.check_serial(prob, 1e-10, 1e-9, kkt_tol=1e-9)
    assert kernel in s._impl.kernel_name and tuple(s._impl.dims[0][:2]) == (nx, nu) and s._impl.padded
    par = pc.check_parallel(prob, 1e-10, legs, 1e-8, rounds=1, rng=rng)
    assert par._impl.kernel_name.startswith("wave_leg<")


@pytest.mark.parametrize("mode,horz,tol", [("W", 100, 1e-9), ("F", 50, 1e-6)])
def test_dense_solver_random_large_problem(mode, horz, tol):   # tests/gar/riccati.cpp:141-155
    nx, nu = 36, 12
    prob = synth.generate_lq_problem(42, np.zeros(nx), horz, nx, nu, mode=mode)
    pc.check_dense(prob, 1e-14, tol, kkt_tol=1e-8)


@pytest.mark.parametrize("nx,nu,nc,nth,horz,mu", [(36, 12, 32, 0, 30, 1e-5), (16, 8, 5, 0, 40, 1e-6),
                                                  (10, 4, 0, 2, 100, 1e-12), (7, 3, 2, 3, 25, 1e-6)])
def test_dense_solver_constrained_and_parametric(nx, nu, nc, nth, horz, mu):
    rng = np.random.default_rng(nx + nc)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), horz, nx, nu, nth=nth, nc=nc, mode="W")
    theta = rng.uniform(-1, 1, nth) if nth else None
    pc.check_dense(prob, mu, 1e-8, theta=theta, kkt_tol=1e-6)   # multipliers O(1/mu)


def test_dense_solver_batch_and_cycle_append():
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, horz, mu = 36, 12, 24, 1e-12
    probs = [synth.generate_lq_problem(900 + i, np.ones(nx), horz, nx, nu, mode="W") for i in range(40)]
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], probs[0].nc0, batch=len(probs), dense=True)
    assert s.kernel_name == "dense"
    s.upload(probs)
    assert s.backward(mu) and s.forward()
    for b in (0, 17, 39):
        _, _, ref = pc.oracle_serial(probs[b], mu)
        for A, B in zip(s.solution(b), ref):
            assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)
    # cycleAppend (dense-riccati.hxx:118-146): records rotate on the device, the solver stays usable
    s.cycle_append(probs[0].stages[0].dims)
    for b, p in enumerate(probs):      # the caller writes the new last-but-one knot (here: the old first)
        s.upload_knot(b, horz - 1, p.stages[0])
        p.stages[:horz] = p.stages[1:horz] + [p.stages[0]]
    assert s.backward(mu) and s.forward()
    _, _, ref = pc.oracle_serial(probs[5], mu)
    for A, B in zip(s.solution(5), ref):
        assert pc.maxdiff(A, B) <= 1e-9 * pc.scale_of(ref)


def test_config0_lqr_plumbing_shape():
    """BASELINE.json configs[0] (tests/lqr.cpp:30-57, bench/lqr.cpp): random dense LQR, nx=4,
    nu=2, A = I with a random lower-right 2x2, B ~ N(0,1), Q = M^T M, R = M^T M, terminal cost 10 Q,
    mt19937_64{42}-style seed; pass = KKT <= 1e-9 and agreement with the oracle."""
    rng = np.random.default_rng(42)
    nx, nu, horz = 4, 2, 50
    A = np.eye(nx)
    A[2:, 2:] = rng.standard_normal((2, 2))
    B = rng.standard_normal((nx, nu))
    M = rng.standard_normal((nx, nx))
    Q = M.T @ M
    M = rng.standard_normal((nu, nu))
    R = M.T @ M
    prob = _lq_from_blocks(A, B, np.zeros(nx), Q, R, rng.standard_normal(nx), np.zeros(nu), 10 * Q,
                           rng.standard_normal(nx), horz)
    pc.check_serial(prob, 1e-12, 1e-9, kkt_tol=1e-9)


def test_config4_talos_lq_shape():
    """BASELINE.json configs[4] needs Pinocchio (absent): its LQ sub-problem SHAPE instead --
    bench/lqr.cpp:25-57 with dim = 56, nu = 22 (A = I, B = [I; 0], c = 0.1, w_x = I with w_x(0,0) = 2,
    w_u = 1e-2 I), N = 275 -- serial in time on the generic kernels (188 KB of LDS before the
    value-function buffers were shared, 162 KB now)."""
    nx, nu, horz = 56, 22, 275
    A = np.eye(nx)
    B = np.eye(nx, nu)
    wx = np.eye(nx)
    wx[0, 0] = 2.0
    rng = np.random.default_rng(7)
    prob = _lq_from_blocks(A, B, np.full(nx, 0.1), wx, 1e-2 * np.eye(nu), np.zeros(nx), np.zeros(nu), wx,
                           rng.uniform(-1, 1, nx), horz)
    solver, _, _ = pc.check_serial(prob, 1e-10, 1e-8, kkt_tol=1e-8)
    # round 2: off the generic kernels -- the controls are padded to 24 and the backward sweep runs on the
    # one-wave-per-problem kernel (five tile columns; forward and initial stage: generic kernels)
    assert solver.kernel_name == "pair<56,24>"
    # the reference's own generators on this shape, every factor block; both kernels of the wide family
    import os
    for variant, name in (("pair", "pair<56,24>"), ("single", "wave<56,24>")):
        os.environ["GAR_HIP_WIDE"] = variant
        try:
            for mode, tol in (("W", 1e-9), ("F", 1e-6)):
                p2 = synth.generate_lq_problem(5600, np.ones(nx), 40, nx, nu, mode=mode)
                s2, _, _ = pc.check_serial(p2, 1e-12, tol, kkt_tol=1e-6 if mode == "F" else 1e-9)
                assert s2.kernel_name == name
        finally:
            del os.environ["GAR_HIP_WIDE"]
    # PARALLEL mode, as the reference benchmarks this shape (bench/talos-walk.cpp:102-127, bench/lqr.cpp:112-134:
    # LQSolverChoice::PARALLEL with 2-8 threads).  Round 3: segment legs (gar_leg_seg.hpp) -- the two-wave stage kernel
    # over each leg's stages, the parameter part by the generic matrix recursion; before: refused (the generic
    # leg kernels need 270 KB of LDS at nth = 56).  Full factors, K0 after collapseFeedback, at N = 275.
    for legs in (2, 5, 8, 16):
        par = pc.check_parallel(prob, 1e-10, legs, 1e-8)
        assert par._impl.kernel_name == "pair_leg<56,24>"
        # from 4 legs on the reduced condensed system is solved by block cyclic reduction (gar_condensed_cr.hpp)
        assert par._impl.condensed_solver_name == ("reduced+cyclic" if legs >= 4 else "reduced+chain")
    p2 = synth.generate_lq_problem(5601, np.ones(nx), 60, nx, nu, mode="F")
    par = pc.check_parallel(p2, 1e-10, 4, 1e-6)
    assert par._impl.kernel_name == "pair_leg<56,24>"
    # and when the segment-leg family is switched off the generic leg kernels' need is still refused loudly
    import os
    os.environ["GAR_HIP_SEG_LEGS"] = "0"
    try:
        from aligator_amd.gar import ParallelRiccatiSolver
        with pytest.raises(RuntimeError, match="LDS"):
            ParallelRiccatiSolver(prob.copy(), 5)
    finally:
        del os.environ["GAR_HIP_SEG_LEGS"]

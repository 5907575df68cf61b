"""The pipelined sweep (gar_hip_set_pipeline, csrc/gar_forward_lean.hpp): the forward sweep of one half of the batch
beside the backward sweep of the other half.  It must change NOTHING in the results: every test here compares it bit
for bit with the plain call sequence on the same solver, and the plain sequence is what the parity tests pin to the
oracle and to the reference's compiled outputs (test_gpu_parity.py, test_ref_headline.py).

CPU: the unmodified sources on the lane-per-thread emulator (index maps, LDS images of the DMA pieces, the split of
the batch, the deferred initial stage; the emulator's copies complete at once, so the s_waitcnt arithmetic is NOT
exercised there).  GPU (-m gpu): the same comparisons through the real library, where it is."""
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import synth
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    return EMU


def _problems(nx, nu, N, batch, mode="W", seed=700, x0=None):
    return [synth.generate_lq_problem(seed + i, np.zeros(nx) if x0 is None else x0(i), N, nx, nu, mode=mode)
            for i in range(batch)]


def _sweep(s, mu, pipelined, steps=1):
    s.set_pipeline(2 if pipelined else 0)
    for _ in range(steps):
        s.backward_async(mu)
        s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    sols = [[a.copy() for a in s.solution(b)] for b in range(s.batch)]
    gains = [[(f.copy(), k.copy()) for f, k in zip(*s.gains_all(b))] for b in range(s.batch)]
    inits = [tuple(np.array(a).copy() for a in s.initial(b)[:1]) for b in range(s.batch)]
    return sols, gains, inits


def _same(a, b):
    for pa, pb in zip(a[0], b[0]):                    # solutions: xs, us, vs, lbdas per problem
        for A, B in zip(pa, pb):
            for x, y in zip(A, B):
                assert pc.maxdiff(x, y) == 0.0
    for pa, pb in zip(a[1], b[1]):                    # every stage's ff / fb
        for (f1, k1), (f2, k2) in zip(pa, pb):
            assert pc.maxdiff(f1, f2) == 0.0 and pc.maxdiff(k1, k2) == 0.0
    for ia, ib in zip(a[2], b[2]):
        for x, y in zip(ia, ib):
            assert pc.maxdiff(x, y) == 0.0


def _check_shape(lib, nx, nu, N, batch, mode="W", steps=1):
    from aligator_amd.gar import BatchedRiccatiSolver
    probs = _problems(nx, nu, N, batch, mode, x0=lambda i: np.linspace(-1, 1, nx) * (i + 1))
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch, lib_path=lib)
    assert s.kernel_name == f"wave<{nx},{nu}>"
    s.upload(probs)
    plain = _sweep(s, 1e-12, False)
    piped = _sweep(s, 1e-12, True, steps=steps)
    assert s.pipeline == 2
    _same(plain, piped)
    again = _sweep(s, 1e-12, False)                   # and back: the plain path behind a pipelined one
    _same(plain, again)
    # against the oracle as well (one problem): the comparison above is not two copies of one mistake
    from oracle import oracle as ora
    op = ora.Problem.from_knots(probs[-1].stages, probs[-1].G0, probs[-1].g0)
    osol = ora.ProximalRiccatiSolver(op)
    osol.backward(1e-12)
    from aligator_amd.gar import lqrInitializeSolution
    ref = lqrInitializeSolution(probs[-1])
    osol.forward(*ref)
    scale = max(float(np.max(np.abs(a))) for A in ref for a in A if a.size)
    err = max(pc.maxdiff(a, b) for A, B in zip(piped[0][-1], ref) for a, b in zip(A, B) if a.size)
    assert err <= pc.TOL[mode] * max(1.0, scale)
    s.close()


@pytest.mark.parametrize("nx,nu,N,batch", [(36, 12, 5, 5), (8, 4, 7, 2), (16, 8, 4, 3), (12, 4, 1, 4), (32, 12, 3, 2)])
def test_pipelined_sweep_is_bitwise_the_plain_sweep(emu, nx, nu, N, batch):
    _check_shape(emu, nx, nu, N, batch, steps=2)


def test_pipelined_sweep_generator_F(emu):
    _check_shape(emu, 36, 12, 4, 3, mode="F")


def test_pipelined_sweep_defers_a_general_initial_stage(emu):
    """G0 not +-I: no closed form, and the pipelined launch carries no LDS for the fused kkt0 -- the sweep flags the
    problem and gar_initial_wave (launched behind it on the flagged problems only) factorises [Vxx0 G0^T; G0 0] as the
    unfused path does.  One problem of the batch keeps G0 = -I: both forms in one launch."""
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N, batch = 8, 4, 6, 4
    rng = np.random.default_rng(3)
    probs = _problems(nx, nu, N, batch)
    for p in probs[1:]:
        p.G0[...] = -np.eye(nx) + 0.3 * rng.standard_normal((nx, nx))
        p.g0[...] = rng.standard_normal(nx)
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch, lib_path=emu)
    s.upload(probs)
    plain = _sweep(s, 1e-12, False)
    piped = _sweep(s, 1e-12, True)
    for pa, pb in zip(plain[0], piped[0]):
        for A, B in zip(pa, pb):
            for x, y in zip(A, B):
                assert pc.maxdiff(x, y) <= 1e-12 * max(1.0, float(np.max(np.abs(x))) if x.size else 1.0)
    from aligator_amd.gar import lqrComputeKktError
    for b, p in enumerate(probs):
        assert max(lqrComputeKktError(p, *piped[0][b], mueq=1e-12)) <= 1e-9
    s.close()


def test_pipelined_sweep_reports_failed_factorisations(emu):
    """a singular Rhat in ONE problem of the second half: that problem fails, the others do not"""
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N, batch = 8, 4, 3, 4
    probs = _problems(nx, nu, N, batch)
    bad = probs[3]
    for k in bad.stages[:-1]:
        k.R[...] = 0.0
        k.S[...] = 0.0
        k.B[...] = 0.0
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch, lib_path=emu)
    s.upload(probs)
    s.set_pipeline(2)
    s.backward_async(1e-12)
    s.forward_async()
    s.sync()
    assert s.num_failed() == 1
    s.close()


def test_pipelined_sweep_on_the_mpc_ring(emu):
    """cycleAppend under the pipelined schedule: the records never move, logical stage t lives in slot (t + ring0)
    mod N -- in gar_forward_lean's DMA addresses as in the backward sweep's -- through more cycles than stages, against
    the oracle on the caller's rotated problems (tests/mpc-cycle.cpp; proximal-riccati.hxx:79-86)"""
    s = pc.check_cycle_append_ring(emu, nx=8, nu=4, horz=4, cycles=6, family="wave", pipeline=True)
    assert s.kernel_name == "wave<8,4>" and s.pipeline == 2


def test_dims_changing_cycle_append_under_the_pipeline(emu):
    """A cycleAppend whose knot has OTHER dimensions rebuilds layout and kernel family (gar_hip_cycle_append's rebuild
    path): the pipelined schedule must not keep launching the old shape's half-batch kernels over the new records --
    it is re-validated against the new family (off here: the mixed-dimension problem runs on the any-dimension
    kernels) and the async call pair gives the oracle's solution.  Then a second rebuild back to a uniform shape, where
    the pipeline the caller asked for comes back on."""
    from aligator_amd.gar import BatchedRiccatiSolver, lqrInitializeSolution
    from oracle import oracle as ora
    nx, nu, N, batch, mu = 8, 4, 5, 3, 1e-10
    rng = np.random.default_rng(5)
    probs = [synth.generate_lq_problem(rng, rng.standard_normal(nx), N, nx, nu, mode="W") for _ in range(batch)]
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch, lib_path=emu)
    s.upload(probs)
    s.set_pipeline(2)
    s.backward_async(mu)
    s.forward_async()
    s.sync()
    assert s.pipeline == 2 and s.kernel_name == "wave<8,4>"

    def check(expect_pipeline, expect_kernel):
        s.upload(probs)                     # a rebuild does not carry resident problem data over
        s.backward_async(mu)
        s.forward_async()
        s.sync()
        assert s.num_failed() == 0
        assert s.pipeline == expect_pipeline and s.kernel_name == expect_kernel, (s.pipeline, s.kernel_name)
        for b, p in enumerate(probs):
            op = ora.Problem.from_knots(p.stages, p.G0, p.g0)
            osol = ora.ProximalRiccatiSolver(op)
            osol.backward(mu)
            ref = lqrInitializeSolution(p)
            osol.forward(*ref)
            scale = max(1.0, max(float(np.max(np.abs(a))) for A in ref for a in A if a.size))
            for A, B in zip(s.solution(b), ref):
                assert pc.maxdiff(A, B) <= 1e-9 * scale, b

    odd = synth.generate_knot(rng, nx, 3, mode="W")            # nu = 3 on the new last-but-one knot
    s.cycle_append(odd.dims)
    for p in probs:
        k = synth.generate_knot(rng, nx, 3, mode="W")
        p.stages[:N] = p.stages[1:N] + [k]
    check(0, "generic")
    for _ in range(N):                                          # ... until every knot is (8, 4) again
        s.cycle_append(probs[0].stages[0].dims if probs[0].stages[0].dims[1] == nu else (nx, nu, 0, nx, 0))
        for p in probs:
            p.stages[:N] = p.stages[1:N] + [synth.generate_knot(rng, nx, nu, mode="W")]
    check(2, "wave<8,4>")
    s.close()


def test_the_library_chooses_its_schedule(emu):
    """gar_hip_set_pipeline(s, -1) (round 6: what a NEW solver starts with, switch PIPELINE): never an error; plain
    below 8 x #CUs problems or where the family does not apply; the switch forces either schedule at create."""
    from aligator_amd import _lib
    from aligator_amd.gar import BatchedRiccatiSolver
    L = _lib.load(emu)
    dims = [(8, 4, 0, 8, 0)] * 3 + [(8, 0, 0, 8, 0)]
    s = BatchedRiccatiSolver(dims, 8, batch=4, lib_path=emu)
    assert s.pipeline == 0                      # 4 problems do not fill a chip twice
    s.set_pipeline(-1)
    assert s.pipeline == 0
    s.set_pipeline(2)
    assert s.pipeline == 2
    s.set_pipeline(-1)                          # back to the library's choice
    assert s.pipeline == 0
    s.close()
    try:
        assert L.gar_hip_set_option(b"PIPELINE", b"2") == 0
        s = BatchedRiccatiSolver(dims, 8, batch=4, lib_path=emu)
        assert s.pipeline == 2
        s.close()
        legs = BatchedRiccatiSolver(dims, 8, batch=4, num_legs=2, lib_path=emu)   # not the family: plain, no error
        assert legs.pipeline == 0
        legs.set_pipeline(-1)
        assert legs.pipeline == 0
        legs.close()
    finally:
        L.gar_hip_set_option(b"PIPELINE", None)


def test_pipeline_is_refused_where_it_does_not_apply(emu):
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N = 8, 4, 8
    dims = [k.dims for k in _problems(nx, nu, N, 1)[0].stages]
    one = BatchedRiccatiSolver(dims, nx, batch=1, lib_path=emu)
    with pytest.raises(RuntimeError, match="pipelined sweep"):
        one.set_pipeline(2)
    one.close()
    legs = BatchedRiccatiSolver(dims, nx, batch=4, num_legs=2, lib_path=emu)
    with pytest.raises(RuntimeError, match="pipelined sweep"):
        legs.set_pipeline(2)
    legs.close()
    ok = BatchedRiccatiSolver(dims, nx, batch=4, lib_path=emu)
    with pytest.raises(RuntimeError):
        ok.set_pipeline(3)
    ok.set_pipeline(2)
    ok.set_pipeline(0)
    assert ok.pipeline == 0
    ok.close()


# ---- the real library -------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nx,nu,N,batch,mode,steps", [(36, 12, 64, 515, "W", 3), (36, 12, 33, 300, "F", 2),
                                                      (8, 4, 40, 301, "W", 2), (16, 8, 17, 258, "W", 2)])
def test_gpu_pipelined_sweep_is_bitwise_the_plain_sweep(nx, nu, N, batch, mode, steps):
    """batch > 256 CUs (the one-wave-per-problem family), odd sizes: uneven halves, a partial last forward workgroup.
    Here the DMA pieces are asynchronous: the vmcnt arithmetic of gar_forward_lean is what is being tested."""
    _check_shape(None, nx, nu, N, batch, mode=mode, steps=steps)


@pytest.mark.gpu
def test_gpu_pipelined_sweep_defers_a_general_initial_stage():
    """on the device: G0 != +-I in most problems of a batch larger than the chip (the pipelined launch carries no LDS
    for the fused kkt0: flagged per problem, factorised by gar_initial_wave behind the sweep), against the plain
    schedule's fused initial stage and the KKT residual of every 37th problem"""
    from aligator_amd.gar import BatchedRiccatiSolver, lqrComputeKktError
    nx, nu, N, batch = 36, 12, 24, 300
    rng = np.random.default_rng(3)
    probs = _problems(nx, nu, N, batch)
    for i, p in enumerate(probs):
        if i % 5:
            p.G0[...] = -np.eye(nx) + 0.2 * rng.standard_normal((nx, nx))
            p.g0[...] = rng.standard_normal(nx)
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=batch)
    assert s.kernel_name == "wave<36,12>"
    s.upload(probs)
    s.set_pipeline(0)
    s.backward_async(1e-12); s.forward_async(); s.sync()
    plain = [s.solution(b) for b in range(0, batch, 37)]
    s.set_pipeline(2)
    for _ in range(2):
        s.backward_async(1e-12); s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    for k, b in enumerate(range(0, batch, 37)):
        sol = s.solution(b)
        for A, B in zip(sol, plain[k]):
            for x, y in zip(A, B):
                assert pc.maxdiff([x], [y]) <= 1e-11 * max(1.0, float(np.abs(y).max()) if y.size else 1.0)
        scale = max(1.0, max(float(np.abs(v).max()) for part in sol for v in part if v.size))
        assert max(lqrComputeKktError(probs[b], *sol, mueq=1e-12)) <= 1e-9 * scale   # (a perturbed G0: |lbd| ~ 1e2-1e3)
    s.close()


@pytest.mark.gpu
def test_gpu_pipelined_headline_batch_against_the_plain_sweep():
    """the headline launch geometry: 4 096 problems of N = 256 generated on the device, five pipelined steps in a row
    (forward sweeps riding beside the next step's backward sweeps), then bit for bit against one plain step"""
    import torch
    from aligator_amd import synth_device
    from aligator_amd.gar import BatchedRiccatiSolver
    nx, nu, N, batch = 36, 12, 256, 4096
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=batch)
    synth_device.fill_problems(s, seed=5, mode="W", keep=())
    s.backward_async(1e-14)
    s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    nd = s._L.gar_hip_solution_doubles(s.handle)
    sol = torch.empty(batch * nd, dtype=torch.float64, device="cuda")
    import ctypes as C
    src = s._L.gar_hip_device_solutions(s.handle)
    torch.cuda.synchronize()
    # (device-to-device copy of the solution records through the HIP runtime torch already loaded)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(sol.data_ptr()), C.c_void_p(src), C.c_size_t(batch * nd * 8), 3) == 0
    hip.hipMemset(C.c_void_p(src), 0, C.c_size_t(batch * nd * 8))
    s.set_pipeline(2)
    for _ in range(5):
        s.backward_async(1e-14)
        s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    piped = torch.empty_like(sol)
    assert hip.hipMemcpy(C.c_void_p(piped.data_ptr()), C.c_void_p(s._L.gar_hip_device_solutions(s.handle)),
                         C.c_size_t(batch * nd * 8), 3) == 0
    assert torch.equal(sol, piped)
    assert float(sol.abs().max()) > 0
    s.close()


@pytest.mark.gpu
def test_gpu_the_default_schedule_is_the_librarys_choice():
    """What a caller gets WITHOUT asking (round 6): a new solver of the headline family starts pipelined once each half of
    its batch fills every SIMD (batch >= 8 x #CUs), plain below; the default's results are bit for bit those of the plain
    schedule on the same data (short horizon: the test is about the choice, not the sweep)."""
    import torch
    from aligator_amd import synth_device
    from aligator_amd.gar import BatchedRiccatiSolver
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    nx, nu, N = 36, 12, 8
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    small = BatchedRiccatiSolver(dims, nx, batch=8 * cus - 1)
    assert small.pipeline == 0 and small.kernel_name == "wave<36,12>"
    small.close()
    s = BatchedRiccatiSolver(dims, nx, batch=8 * cus)
    assert s.pipeline == 2
    synth_device.fill_problems(s, seed=3, mode="W")
    piped = _sweep(s, 1e-12, True, steps=2)
    s.set_pipeline(-1)
    assert s.pipeline == 2
    plain = _sweep(s, 1e-12, False)
    _same(plain, piped)
    s.close()

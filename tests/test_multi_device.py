"""ONE process, SEVERAL devices (include/gar_hip.h, gar_hip_multi_create; csrc/gar_multi.hpp): the horizon of one
problem sharded over W devices behind the one solver object `SolverProxDDP` holds -- the multi-device form of
gar::ParallelRiccatiSolver (gar/parallel-solver.hxx:132-243), with the boundary exchange inside backward().

CPU part: the kernel + C-ABI sources on the wave emulator with W in {2, 3} VIRTUAL devices -- against the serial
oracle, the reference's own outputs (tests/golden/ref/parallel_shape_nx8_N17.npz) and, bitwise, the one-device solver
with the same legs.  GPU part (-m gpu): W ranked solvers sharing the one GPU of the box (device-to-device exchange,
both the gather kernel and the copy path), at N = 2048, nx = 36 -- BASELINE configs[3]'s shape."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver, ParallelRiccatiSolver, lqrInitializeSolution
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    lib = C.CDLL(EMU)
    lib.emu_set_device_count(3)   # three virtual devices
    yield EMU
    lib.emu_set_device_count(1)


def _flat(sol):
    return np.concatenate([np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0) for part in sol])


@pytest.mark.parametrize("devices,legs,horz,nx,nu,pad", [
    ([0, 1], 2, 11, 8, 4, "1"),       # wave_leg<8,4>, one leg per device
    ([0, 1, 2], 5, 17, 8, 4, "1"),    # uneven: 1 + 2 + 2 legs
    ([0, 1, 2], 3, 13, 6, 3, "1"),    # padded onto (8, 4) inside the C ABI
    ([0, 1], 4, 13, 6, 3, "0"),       # the any-dimension leg kernels
    ([2, 0, 1], 4, 12, 12, 6, "1"),   # devices in any order; (12, 6) -> wave_leg<12,8>
])
def test_multi_device_equals_the_serial_oracle(emu, monkeypatch, devices, legs, horz, nx, nu, pad):
    """tests/gar/parallel.cpp:185-245 with the legs on W devices: solution vs the serial oracle, every stage's
    factors and the collapsed K0 vs the oracle's leg-parallel solver, a second sweep on modified data."""
    monkeypatch.setenv("GAR_HIP_PAD", pad)
    rng = np.random.default_rng(17)
    prob = synth.generate_lq_problem(rng, np.zeros(nx), horz, nx, nu)
    par = pc.check_parallel(prob, 1e-9, legs, 1e-7, emu, rounds=1, rng=rng, devices=devices)
    L, h = par._impl._L, par._impl.handle
    assert L.gar_hip_num_devices(h) == len(devices)
    assert L.gar_hip_multi_exchange_name(h) == b"pull"
    owners = [L.gar_hip_stage_device(h, t) for t in range(horz + 1)]
    assert owners[0] == devices[0] and owners[-1] == devices[-1] and len(set(owners)) == len(set(devices))


def test_multi_device_constrained_knots(emu):
    """nc > 0 on every knot in leg mode (folded onto the unconstrained wave-leg family, gar_fold.hpp) over 2 devices."""
    rng = np.random.default_rng(3)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(8), 11, 8, 4, nc=4, mode="W")
    pc.check_parallel(prob, 1e-6, 3, 1e-7, emu, devices=[0, 1], conditioned=True)


@pytest.mark.parametrize("legs,devices", [(3, [0, 1]), (5, [0, 1, 2])])
def test_multi_device_coupled_constraints(emu, legs, devices):
    """D != 0 in leg mode over several devices: every sub-solver runs the constrained segment legs (gar_cstr_seg.hpp) on
    ITS legs (leg_begin > 0 on all but the first) -- plain part, leg-end kernel, parameter recursion, roll-out -- and the
    boundary tuples travel as for every other family."""
    rng = np.random.default_rng(4)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(8), 16, 8, 4, nc=4, mode="W")
    for t, k in enumerate(prob.stages[:-1]):
        k.C[...] = rng.uniform(-1, 1, k.C.shape)
        k.D[...] = rng.uniform(-1, 1, k.D.shape)
        if t % 4 == 0:
            k.R[...] *= 1e-2                      # (knots on which Bunch-Kaufman pivots, inside legs too)
    par = pc.check_parallel(prob, 1e-6, legs, 1e-7, emu, devices=devices, conditioned=True)
    assert "wave_seg<8,4,4>" in par._impl.kernel_name


@pytest.mark.parametrize("exchange", ["pull", "copy", "nopeer"])
def test_multi_device_matches_the_reference_outputs(emu, monkeypatch, exchange):
    """The reference's own ParallelRiccatiSolver outputs (tests/golden/ref, compiled from /root/reference by
    tests/golden/make_ref_golden.py): solution, every stage's factors, the collapsed K0 -- with the legs on 2 and 3
    devices, through each form of the exchange."""
    from test_golden import _ref_pair, _unflat, assert_factors_match_reference, assert_matches, load_fixture, ref_tol, _rel
    if exchange == "copy":
        monkeypatch.setenv("GAR_HIP_MULTI_EXCHANGE", "copy")
    if exchange == "nopeer":
        monkeypatch.setenv("GAR_EMU_NO_PEER", "1")   # hipDeviceCanAccessPeer says no: hipMemcpyPeerAsync
    path = os.path.join(HERE, "golden", "ref", "parallel_shape_nx8_N17.npz")
    src, z = _ref_pair(path)
    prob, mueq, _, _ = load_fixture(src)
    tol = max(ref_tol(path), 1e-9)
    for J in sorted({int(k[3:k.index("_")]) for k in z.files if k.startswith("par")}):
        for W in (2, 3):
            if W > J:
                continue
            par = ParallelRiccatiSolver(prob.copy(), J, lib_path=emu, devices=list(range(W)))
            assert par._impl._L.gar_hip_multi_exchange_name(par._impl.handle).decode() == ("pull" if exchange == "pull" else "copy")
            par.maxRefinementSteps = 10
            par.backward(mueq)
            psol = lqrInitializeSolution(prob)
            par.forward(*psol)
            assert_matches(psol, _unflat(prob, z, f"par{J}_"), tol)
            assert_factors_match_reference(lambda t: par.datas[t], z, prob.horizon, tol, prefix=f"par{J}_s")
            par.collapseFeedback()
            assert _rel(par.getFeedback(0), z[f"par{J}_K0_collapsed"]) <= tol


def _bitwise_against_one_device(lib, prob, legs, devices, mueq, batch=2):
    dims = [k.dims for k in prob.stages]
    one = BatchedRiccatiSolver(dims, prob.nc0, batch=batch, num_legs=legs, lib_path=lib)
    many = BatchedRiccatiSolver(dims, prob.nc0, batch=batch, num_legs=legs, lib_path=lib, devices=devices)
    assert many.kernel_name == one.kernel_name
    rng = np.random.default_rng(5)
    probs = [prob]
    for _ in range(batch - 1):
        p = prob.copy()
        synth.randomly_modify_problem(rng, p)
        probs.append(p)
    for s in (one, many):
        s.upload(probs)
        assert s.backward(mueq) and s.forward()
        s.collapse_feedback()
    N = prob.horizon
    for b in range(batch):
        assert np.array_equal(_flat(one.solution(b)), _flat(many.solution(b)))
        for t in range(N + 1):
            f, g = one.factor(t, b), many.factor(t, b)
            for name in ("ff", "fb", "fth"):
                assert np.array_equal(getattr(f, name), getattr(g, name)), (b, t, name)
            for name in ("Vxx", "vx", "Vxt", "Vtt", "vt"):
                assert np.array_equal(getattr(f.vm, name), getattr(g.vm, name)), (b, t, name)
        for a, c in zip(one.initial(b), many.initial(b)):
            assert np.array_equal(a, c)
        # bulk read-back: every device copies its own stages into the one pinned buffer
        (ra, fa, ba), (rm, fm, bm) = one.fetch_results(b), many.fetch_results(b)
        assert np.array_equal(ra, rm) and np.array_equal(fa, fm) and np.array_equal(ba, bm)
    # what the devices hold is what was uploaded (each device: its own stages + G0 | g0)
    assert np.array_equal(one.download_packed(), many.download_packed())
    return one, many


@pytest.mark.parametrize("nx,nu,legs,devices,pad", [(8, 4, 5, [0, 1, 2], "1"), (6, 3, 4, [0, 1], "1"), (5, 2, 3, [0, 1, 2], "0")])
def test_multi_device_is_bitwise_the_one_device_solver(emu, monkeypatch, nx, nu, legs, devices, pad):
    """Same legs, same kernels, same data: splitting the legs over devices changes no bit -- solution, every factor
    block, kkt0, the bulk read-back (gar_hip_fetch_results), the packed download; batch of 2."""
    monkeypatch.setenv("GAR_HIP_PAD", pad)
    rng = np.random.default_rng(23)
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), 14, nx, nu, mode="W")
    _bitwise_against_one_device(emu, prob, legs, devices, 1e-10)


def test_multi_device_cycle_append_and_errors(emu):
    """cycleAppend in leg mode re-initialises (parallel-solver.hxx:246-258) -- on every device; entry points that
    speak device pointers are refused; one device falls back to the plain solver."""
    nx, nu, N = 8, 4, 11
    rng = np.random.default_rng(4)
    prob = synth.generate_lq_problem(rng, np.zeros(nx), N, nx, nu, mode="W")
    par = ParallelRiccatiSolver(prob, 3, lib_path=emu, devices=[0, 1, 2])
    par.backward(1e-10)
    new = synth.generate_lq_problem(rng, np.zeros(nx), 1, nx, nu, mode="W").stages[0]
    prob.stages = prob.stages[1:N] + [new, prob.stages[N]]
    par.cycleAppend(new)
    assert par._impl._L.gar_hip_num_devices(par._impl.handle) == 3
    par.backward(1e-10)
    sol = lqrInitializeSolution(prob)
    par.forward(*sol)
    plain = prob.copy()
    plain.addParameterization(0)
    _, _, ref = pc.oracle_serial(plain, 1e-10)
    sc = pc.scale_of(ref)
    for a, b in zip(sol, ref):
        assert pc.maxdiff(a, b) <= 1e-8 * sc
    L, h = par._impl._L, par._impl.handle
    assert L.gar_hip_device_problems(h) is None and L.gar_hip_device_factors(h) is None
    assert L.gar_hip_set_stream(h, None) == -3
    assert L.gar_hip_upload_packed_device(h, 0, 1, C.c_void_p(16)) == -3
    dims = np.ascontiguousarray(np.array([k.dims for k in prob.stages], dtype=np.int32))
    dims[:, 4] = 0
    ids = (C.c_int * 3)(0, 1, 2)
    p32 = dims.ctypes.data_as(C.POINTER(C.c_int32))
    assert not L.gar_hip_multi_create(3, ids, N, p32, nx, 1, 2)          # fewer legs than devices
    assert not L.gar_hip_multi_create(2, (C.c_int * 2)(0, 7), N, p32, nx, 1, 2)   # no such device
    one = L.gar_hip_multi_create(1, ids, N, p32, nx, 1, 1)                 # one device: the plain serial solver
    assert one and L.gar_hip_num_devices(one) == 1 and L.gar_hip_multi_exchange_name(one) == b""
    L.gar_hip_solver_destroy(one)


# ---- on the GPU: W ranked solvers sharing the box's one device ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["pull", "copy"])
def test_gpu_multi_device_horizon_2048(monkeypatch, exchange):
    """BASELINE configs[3]'s shape, N = 2048, nx = 36, nu = 12: 16 legs over W = 2 sub-solvers on the SAME device
    (the exchange is a device-to-device gather ordered by HIP events across two streams) -- bitwise the one-solver
    result with the same legs, and the serial oracle's solution."""
    if exchange == "copy":
        monkeypatch.setenv("GAR_HIP_MULTI_EXCHANGE", "copy")
    nx, nu, N, legs = 36, 12, 2048, 16
    prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
    one, many = _bitwise_against_one_device(None, prob, legs, [0, 0], 1e-10, batch=1)
    assert many._L.gar_hip_multi_exchange_name(many.handle).decode() == exchange
    assert many.kernel_name.startswith("wave_leg<36,12>")
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    sc = pc.scale_of(ref)
    for a, b in zip(many.solution(0), ref):
        assert pc.maxdiff(a, b) <= 1e-8 * sc
    # repeated sweeps reuse the events (the readers of sweep k gate the writers of sweep k + 1)
    for _ in range(3):
        assert many.backward(1e-10) and many.forward()
    assert np.array_equal(_flat(one.solution(0)), _flat(many.solution(0)))


@pytest.mark.gpu
@pytest.mark.parametrize("W,legs", [(8, 8), (8, 256), (5, 37), (16, 64)])
@pytest.mark.parametrize("exchange", ["pull", "copy"])
def test_gpu_multi_device_eight_way_at_configs3_shape(monkeypatch, W, legs, exchange):
    """The plumbing of the first 8-GPU run, on the one device this box has: BASELINE configs[3]'s exact shape (N = 2048,
    nx = 36, nu = 12) with W = 8 sub-solvers -- ONE 256-stage leg per sub-solver (legs = 8) and 32 legs per sub-solver
    (legs = 256) -- plus an uneven split (5 sub-solvers, 37 legs) and more sub-solvers than a node has GPUs (16); both
    exchange forms.  Bitwise the one-device solver with the same legs (solution and the bulk read-back of every gain),
    the serial oracle's solution, and three sweeps in a row (the readers of sweep k gate the writers of sweep k + 1
    through W x W events).  What this cannot show is xGMI itself: peer access between distinct devices
    (parallel-solver.hxx:23-28, 132-206 are the reference's partition and exchange)."""
    from aligator_amd.gar import BatchedRiccatiSolver
    if exchange == "copy":
        monkeypatch.setenv("GAR_HIP_MULTI_EXCHANGE", "copy")
    nx, nu, N = 36, 12, 2048
    prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
    dims = [k.dims for k in prob.stages]
    many = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, devices=[0] * W)
    assert many._L.gar_hip_num_devices(many.handle) == W
    assert many._L.gar_hip_multi_exchange_name(many.handle).decode() == exchange
    many.upload([prob])
    for _ in range(3):
        assert many.backward(1e-10) and many.forward()
    one = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs)
    one.upload([prob])
    assert one.backward(1e-10) and one.forward()
    assert np.array_equal(_flat(one.solution(0)), _flat(many.solution(0)))
    for x, y in zip(many.fetch_results(0), one.fetch_results(0)):
        assert np.array_equal(x, y)
    _, _, ref = pc.oracle_serial(prob, 1e-10)
    sc = pc.scale_of(ref)
    for a, b in zip(many.solution(0), ref):
        assert pc.maxdiff(a, b) <= 1e-8 * sc
    many.close()
    one.close()


@pytest.mark.gpu
def test_gpu_bench_lines_for_eight_ranks_on_one_device():
    """the two command lines of the 8-GPU runs, on one device: `bench.py --gpus 8 --same-device --backend gloo` (eight
    ranks, the batch axis) and `--mode horizon --single-process --gpus 8 --same-device` (one process, eight sub-solvers):
    each prints ONE well-formed line"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--same-device", "--backend", "gloo", "--batch", "64",
                        "--steps", "2", "--warmup", "1", "--single-generator", "--no-cpu", "--no-extras", "--pmc", "off"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["batch_per_gpu"] == 64
    assert d["parity"]["max_rel_err_vs_oracle"] < 1e-9 and d["parity"]["failed_factorisations"] == 0
    hs = d.get("horizon_sharded")
    assert hs is None or "error" not in hs, hs
    r = subprocess.run([sys.executable, "bench.py", "--mode", "horizon", "--single-process", "--gpus", "8", "--same-device"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    hs = d["horizon_sharded"]
    assert d["n_gpus"] == 8 and hs["devices"] == [0] * 8 and hs["max_rel_diff_vs_serial"] < 1e-9


@pytest.mark.gpu
def test_gpu_multi_device_three_way_uneven_and_padded():
    """W = 3 sub-solvers on the one device, 7 legs (2 + 2 + 3), the Talos shape (56, 22) padded inside the C ABI onto
    the segment-leg family, and a generic shape; batch of 2."""
    rng = np.random.default_rng(11)
    for nx, nu, N, legs in ((56, 22, 70, 7), (10, 3, 40, 7), (7, 5, 23, 4)):
        prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), N, nx, nu, mode="W")
        _, many = _bitwise_against_one_device(None, prob, legs, [0, 0, 0], 1e-10, batch=2)
        _, _, ref = pc.oracle_serial(prob, 1e-10)
        sc = pc.scale_of(ref)
        for a, b in zip(many.solution(0), ref):
            assert pc.maxdiff(a, b) <= 1e-8 * sc


@pytest.mark.gpu
def test_gpu_bench_horizon_single_process():
    """`python bench.py --mode horizon --single-process --gpus 2`: the configs[3] line from ONE process without
    torchrun (two sub-solvers share this box's GPU); more GPUs than visible fails loudly."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "bench.py", "--mode", "horizon", "--single-process", "--gpus", "2", "--same-device"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    hs = d["horizon_sharded"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and hs["devices"] == [0, 0] and hs["exchange"] == "pull"
    assert hs["max_rel_diff_vs_serial"] < 1e-9 and d["horizon_sharded_N16384"]["max_rel_diff_vs_serial"] < 1e-9
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, "bench.py", "--mode", "horizon", "--single-process", "--gpus", str(n + 1)],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_gpu_first_multi_gpu_script():
    """scripts/first_multi_gpu.sh -- the script of the first run on several physical GPUs (peer-access matrix; one
    process with the legs over the devices, pull vs copy, bitwise the one-device solver at configs[3]'s shape; RCCL
    --mode horizon; the batch axis) -- exercised here with every rank / sub-solver on this box's device (on a box with
    several devices it runs for real): every step must report ok."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(["bash", os.path.join(root, "scripts", "first_multi_gpu.sh"), "--quick"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=2400)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and "first_multi_gpu: every step ok" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    assert "multi-device ok" in r.stdout

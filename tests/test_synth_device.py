"""aligator_amd/synth_device.py: the generator that writes bench.py's problems straight into the packed device records
(headline and, round 6, the secondary shapes: constrained nc > 0 with D = 0 or a random D, the padded wide shape).  On
CPU the same records are built in host memory and swept by the emulator build; each problem read back through
gar_hip_download_packed must be the one the kernels saw: its oracle solution is the sweep's."""
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


@pytest.mark.parametrize("nx,nu,nc,N,coupled,mu,kernel", [(8, 4, 4, 4, True, 1e-6, "wave<8,4,4>"), (8, 4, 4, 3, False, 1e-6, "wave<8,4,4>"),
                                                           (36, 12, 32, 2, True, 1e-8, "wave<36,12,32>"), (56, 22, 0, 2, False, 1e-10, "pair<56,24>"),
                                                           (36, 12, 0, 3, False, 1e-12, "wave<36,12>")])
def test_generated_records_are_the_problems_read_back(nx, nu, nc, N, coupled, mu, kernel):
    from oracle import oracle as ora
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    dims = [(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=3, lib_path=EMU)
    assert s.kernel_name == kernel
    synth_device.fill_problems(s, seed=5, mode="W", coupled=coupled, keep=(0, 1, 2))
    assert s.backward(mu) and s.forward()
    for b in range(3):
        prob = synth_device.download_problem(s, b)
        if nc:
            assert np.array_equal(prob.stages[0].C, np.eye(nc, nx))
            assert (np.abs(prob.stages[0].D).max() > 0.5) == coupled
        op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)
        o = ora.ProximalRiccatiSolver(op)
        o.backward(mu)
        ref = op.initialize_solution()
        o.forward(*ref)
        scale = max(1.0, max(float(np.abs(v).max()) for part in ref for v in part if v.size))
        err = max(float(np.abs(a - c).max()) for A, B in zip(s.solution(b), ref) for a, c in zip(A, B) if a.size) / scale
        assert err <= 1e-9, (b, err)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nu,nc,N,coupled,mu", [(36, 12, 32, 16, False, 1e-11), (36, 12, 32, 16, True, 1e-11), (56, 22, 0, 12, False, 1e-10)])
def test_gpu_generated_secondary_shapes_against_the_oracle(nx, nu, nc, N, coupled, mu):
    """bench.py's secondary shapes as it generates them since round 6 (on the device, every problem its own draw): the
    first and the last problem of a batch larger than the CU count, read back and solved by the oracle."""
    from aligator_amd.gar import lqrComputeKktError
    from oracle import oracle as ora
    dims = [(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)]
    s = BatchedRiccatiSolver(dims, nx, batch=300)
    synth_device.fill_problems(s, seed=9, mode="W", coupled=coupled)
    assert s.backward(mu) and s.forward()
    if nc:
        chain = s.constrained_bk_stages()
        assert (chain[0] > 0) == coupled
    for b in (0, 299):
        prob = synth_device.download_problem(s, b)
        op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)
        o = ora.ProximalRiccatiSolver(op)
        o.backward(mu)
        ref = op.initialize_solution()
        o.forward(*ref)
        scale = max(1.0, max(float(np.abs(v).max()) for part in ref for v in part if v.size))
        err = max(float(np.abs(a - c).max()) for A, B in zip(s.solution(b), ref) for a, c in zip(A, B) if a.size) / scale
        assert err <= 1e-9, (b, err)
        assert max(lqrComputeKktError(prob, *s.solution(b), mueq=mu)) / scale <= 1e-9
    s.close()

"""The reference runs backward / forward under ALIGATOR_NOMALLOC_SCOPED (gar/proximal-riccati.hxx:35,
gar/parallel-solver.hxx; tests/nomalloc.cpp): no allocation inside the sweep.  The same contract here, counted:
every device / pinned-host allocation of the library goes through two counted wrappers (csrc/gar_hip.cpp,
gar_dev_malloc / gar_host_malloc; gar_hip_debug_alloc_count) and the count does not move across 100
upload + backward + forward + bulk read-back rounds -- what one Newton iteration of SolverProxDDP does with
linear_solver_ (solver-proxddp.hxx:608-632)."""
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import _lib, synth
from aligator_amd.gar import BatchedRiccatiSolver

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


def _rounds(lib_path, nx, nu, N, legs, devices, rounds, nc=0):
    L = _lib.load(lib_path)
    prob = synth.generate_lq_problem(3, np.zeros(nx), N, nx, nu, nc=nc, mode="W")
    dims = [k.dims for k in prob.stages]
    s = BatchedRiccatiSolver(dims, prob.nc0, batch=1, num_legs=legs, lib_path=lib_path, devices=devices)
    s.upload([prob])
    assert s.backward(1e-8) and s.forward()
    s.fetch_results(0)                     # the pinned read-back buffer is created on first use
    before = L.gar_hip_debug_alloc_count()
    assert before > 0
    for _ in range(rounds):
        for t, k in enumerate(prob.stages):   # the caller's problem is re-read on every backward
            s.upload_knot(0, t, k)
        s.set_init(0, prob.G0, prob.g0)
        assert s.backward(1e-8) and s.forward()
        if legs > 1:
            s.collapse_feedback()
        s.fetch_results(0)
        s.solution(0)
    assert L.gar_hip_debug_alloc_count() == before, (s.kernel_name, L.gar_hip_debug_alloc_count() - before)
    return s.kernel_name


@pytest.mark.parametrize("nx,nu,nc,N,legs,devices", [(8, 4, 0, 9, 1, None), (6, 3, 2, 7, 1, None), (8, 4, 0, 11, 3, None),
                                                     (8, 4, 2, 11, 2, None), (8, 4, 0, 11, 3, [0, 0])])
def test_no_allocation_inside_the_sweep_on_the_emulator(nx, nu, nc, N, legs, devices):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    _rounds(EMU, nx, nu, N, legs, devices, 3, nc=nc)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nu,nc,N,legs,devices", [(36, 12, 0, 64, 1, None), (36, 12, 32, 32, 1, None), (56, 22, 0, 48, 1, None),
                                                     (36, 12, 0, 128, 8, None), (56, 22, 0, 64, 4, None), (7, 3, 2, 20, 3, None),
                                                     (36, 12, 0, 128, 8, [0, 0]), (36, 12, 0, 40, 1, [0])])
def test_no_allocation_inside_the_sweep(nx, nu, nc, N, legs, devices):
    _rounds(None, nx, nu, N, legs, devices, 100, nc=nc)

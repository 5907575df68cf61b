"""Device-resident updateLQSubproblem (the next row either side of the hot path):
the kernel's knots must equal the numpy restatement of solver-proxddp.hxx:734-805 BITWISE
(it only copies and adds, in the reference's order), on the emulator (CPU) and on the GPU;
then the sweep on the device-assembled problem must match the oracle on the host-assembled one."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from aligator_amd.lqr import LqrKnot, LqrProblem, lqrInitializeSolution
from oracle.update_lq import update_lq_subproblem
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


def random_derivs(rng, dims, nc0):
    from aligator_amd.gar import BatchedRiccatiSolver
    derivs = []
    for (nx, nu, nc, nx2, _) in dims:
        d = {}
        for name, shp in BatchedRiccatiSolver.deriv_shapes(nx, nu, nc, nx2).items():
            d[name] = rng.standard_normal(shp)
        for name, n in (("Lxx", nx), ("Luu", nu), ("Hxx", nx), ("Huu", nu)):
            m = rng.standard_normal((n, n + 1))
            d[name] = m @ m.T / max(n, 1) * (0.05 if name.startswith("H") else 1.0)
        d["Jx"] = np.eye(nx2, nx) + 0.1 * rng.standard_normal((nx2, nx))
        derivs.append(d)
    nx0 = dims[0][0]
    init = {"Jx": -np.eye(nc0, nx0), "value": rng.standard_normal(nc0),
            "Hxx": 0.01 * np.eye(nx0)}
    return derivs, init


def run_case(lib_path, dims, nc0, batch, preg, hess_exact, device):
    from aligator_amd.gar import BatchedRiccatiSolver
    rng = np.random.default_rng(77)
    s = BatchedRiccatiSolver(dims, nc0, batch=batch, lib_path=lib_path)
    host_probs, bufs = [], []
    for b in range(batch):
        derivs, init = random_derivs(rng, dims, nc0)
        bufs.append(s.pack_derivs(derivs, init))
        prob = LqrProblem([LqrKnot(*d[:4]) for d in dims], nc0)
        update_lq_subproblem(prob, derivs, init, preg, hess_exact)
        host_probs.append(prob)
    flat = np.concatenate(bufs)
    if device:
        import torch
        dev = torch.from_numpy(flat).cuda()
        torch.cuda.synchronize()
        ptr = dev.data_ptr()
    else:
        ptr = flat.ctypes.data  # the emulator's "device" memory is host memory
    s.update_lq_subproblem_device(ptr, preg, hess_exact)
    s.sync()
    packed = s.download_packed()
    for b, prob in enumerate(host_probs):
        got = packed[b * s.problem_doubles:(b + 1) * s.problem_doubles]
        assert np.array_equal(got, s.pack(prob)), f"problem {b}: device-assembled knots differ"
    # and the sweep on the device-assembled problems matches the oracle
    # (constrained stages: a proximal weight that keeps the KKT system well conditioned)
    mueq = 1e-10 if all(d[2] == 0 for d in dims) else 1e-2
    s.backward(mueq)
    s.forward()
    for b, prob in enumerate(host_probs):
        _, _, ref = pc.oracle_serial(prob, mueq)
        sc = pc.scale_of(ref)
        for A, B in zip(s.solution(b), ref):
            assert pc.maxdiff(A, B) <= 1e-9 * sc


CASES = [([(4, 2, 0, 4, 0)] * 5 + [(4, 0, 0, 4, 0)], 4, 2, 1e-3, True),
         ([(6, 3, 2, 6, 0)] * 3 + [(6, 0, 1, 6, 0)], 6, 1, 0.0, False),
         ([(8, 4, 0, 8, 0)] * 4 + [(8, 0, 0, 8, 0)], 8, 3, 1e-6, True)]


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    return EMU


@pytest.mark.parametrize("pad", ["0", "1"])
@pytest.mark.parametrize("dims,nc0,batch,preg,exact", CASES)
def test_update_lq_on_emulator(emu_lib, monkeypatch, dims, nc0, batch, preg, exact, pad):
    # pad = 1: (4, 2) runs padded onto the (8, 4) family -- the derivative records keep the CALLER's dimensions, the
    # kernel scatters into the padded knots and writes the dummy rows / columns (gar_update_lq_padded)
    monkeypatch.setenv("GAR_HIP_PAD", pad)
    run_case(emu_lib, dims, nc0, batch, preg, exact, device=False)


# shapes the library pads inside the C ABI: BASELINE configs[0] (4, 2), configs[2] (12, 6), the Talos walk's (56, 22)
PADDED = [([(4, 2, 0, 4, 0)] * 5 + [(4, 0, 0, 4, 0)], 4, 2, 1e-3, True, "<8,4>"),
          ([(12, 6, 0, 12, 0)] * 4 + [(12, 0, 0, 12, 0)], 12, 2, 1e-6, True, "<12,8>"),
          ([(10, 3, 0, 10, 0)] * 3 + [(10, 0, 0, 10, 0)], 7, 1, 0.0, False, "<12,4>"),
          ([(56, 22, 0, 56, 0)] * 3 + [(56, 0, 0, 56, 0)], 56, 1, 1e-8, True, "pair<56,24>")]


@pytest.mark.parametrize("dims,nc0,batch,preg,exact,family", PADDED)
def test_update_lq_on_a_padded_solver(emu_lib, dims, nc0, batch, preg, exact, family):
    from aligator_amd.gar import BatchedRiccatiSolver
    s = BatchedRiccatiSolver(dims, nc0, batch=1, lib_path=emu_lib)
    assert s.padded and family in s.kernel_name, s.kernel_name
    run_case(emu_lib, dims, nc0, batch, preg, exact, device=False)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,nc0,batch,preg,exact",
                         CASES + [([(36, 12, 0, 36, 0)] * 16 + [(36, 0, 0, 36, 0)], 36, 4, 1e-8, True)] +
                         [c[:5] for c in PADDED] + [([(56, 22, 0, 56, 0)] * 32 + [(56, 0, 0, 56, 0)], 56, 3, 1e-8, True)])
def test_update_lq_on_gpu(dims, nc0, batch, preg, exact):
    run_case(None, dims, nc0, batch, preg, exact, device=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,nc0,batch,preg,exact", CASES)
def test_update_lq_on_gpu_unpadded(monkeypatch, dims, nc0, batch, preg, exact):
    monkeypatch.setenv("GAR_HIP_PAD", "0")
    run_case(None, dims, nc0, batch, preg, exact, device=True)

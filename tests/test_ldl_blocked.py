"""csrc/gar_ldl_blocked.hpp -- the wave-scope BLOCKED L D L^T (panels on DPP broadcasts, trailing matrix on MFMA tiles).
In production it factorises the 36 x 36 / 32 x 32 blocks of the cyclic reduction (every leg-mode test of the specialised
shapes runs it).  Its two other uses were measured on the MI355X and are compiled out by default -- Rhat 24 x 24 of
pair<56,24> (no gain) and the coupled 44 x 44 reduced KKT matrix, in place on the row-packed triangle (slower: register
spills in a kernel that is full already; profiles/r06_ab_*) -- but they stay correct code paths: this test builds the
kernel sources on the emulator WITH them (SAN carries the two switches into a build directory of its own) and checks
the two stages against the oracle: plain, pivoting (a column that fails Bunch-Kaufman's first test leaves the blocked
routine), the in-place packed policy with a last panel of 8 columns."""
import os
import subprocess

import numpy as np

from aligator_amd import synth
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))


import pytest


@pytest.mark.parametrize("build,switches", [("blocked", "-DGAR_PAIR_BLOCKED_LDL=1 -DGAR_COUPLED_BLOCKED_LDL=1"),
                                            # the hybrid of the coupled stage: Rhat's columns as ONE panel (NPANELS = 1), the
                                            # Schur complement's factorisation in registers from column NU on (K0 = NU)
                                            ("hybrid", "-DGAR_COUPLED_HYBRID_LDL=1 -DGAR_PAIR_REFRESH_LANE=0 -DGAR_PAIR_UNEVEN=0")])
def test_blocked_ldl_in_the_pair_and_coupled_stages(build, switches):
    LIB = os.path.join(HERE, "emu", "_build", build, "libgar_hip_emu.so")
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu"), f"BUILD=_build/{build}",
                    f"SAN={switches}", f"_build/{build}/libgar_hip_emu.so"], check=True)
    from aligator_amd.gar import BatchedRiccatiSolver
    from oracle import oracle as ora
    rng = np.random.default_rng(3)
    cases = []
    # pair<56,24>: generator W (first test holds), F (some columns fail it: spd-accept keeps the blocked result)
    for mode in ("W", "F"):
        cases.append((synth.generate_lq_problem(rng, rng.standard_normal(56), 3, 56, 24, mode=mode), 1e-10, "pair<56,24>"))
    # coupled stages: NK = 44 (panels 12 + 12 + 12 + 8, in place, packed) and NK = 8 (one short panel)
    for nx, nu, nc, kern in ((36, 12, 32, "wave<36,12,32>"), (8, 4, 4, "wave<8,4,4>")):
        p = synth.generate_lq_problem(rng, rng.standard_normal(nx), 3, nx, nu, nc=nc, mode="W")
        for k in p.stages[:-1]:
            k.D[...] = rng.uniform(-1, 1, k.D.shape)
        cases.append((p, 1e-7, kern))
    for prob, mu, kern in cases:
        s = BatchedRiccatiSolver([k.dims for k in prob.stages], prob.nc0, batch=1, lib_path=LIB)
        assert s.kernel_name == kern
        s.upload([prob])
        assert s.backward(mu) and s.forward()
        op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)
        o = ora.ProximalRiccatiSolver(op)
        o.backward(mu)
        ref = op.initialize_solution()
        o.forward(*ref)
        scale = max(1.0, max(float(np.abs(v).max()) for part in ref for v in part if v.size))
        err = max(pc.maxdiff(a, b) for A, B in zip(s.solution(0), ref) for a, b in zip(A, B) if a.size) / scale
        assert err <= 1e-9, (kern, err)
        s.close()

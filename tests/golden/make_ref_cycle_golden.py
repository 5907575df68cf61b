#!/usr/bin/env python3
"""Generate tests/golden/ref_cycle/*.npz: the REFERENCE'S OWN code through MPC cycles (rows a10 / f4 of SURVEY 8).

For some of the problems of tests/golden/*.npz: the knots are rotated the way WorkspaceTpl::cycleAppend rotates them
(solvers/proxddp/workspace.hxx:122-126: rotate_vec_left(stages, 0, 1), a new knot in the last-but-one slot), the
reference's ProximalRiccatiSolver::cycleAppend (gar/proximal-riccati.hxx:79-86) is called as solver-proxddp.hxx:208
calls it, then backward + forward -- more cycles than the horizon has stages.  The reference's sources are compiled
UNCHANGED from /root/reference over oracle/ref_shim (oracle/ref_build.sh); /root/reference does not exist on the GPU
box, so the vectors are committed.  Every file holds the new knot of every cycle (all 16 blocks) and, after every
cycle, the reference's solution; after the last two (the ring has wrapped) every stage's ff / fb / Vxx / vx.
What they pin: the oracle on the rotated problem, the kernel sources' ring (gar_hip_cycle_append) on the emulator and
the HIP path on the GPU (tests/test_golden.py::test_*cycle*_reference_outputs).

Run from the repository root (in the build container):  python tests/golden/make_ref_cycle_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from aligator_amd import synth                       # noqa: E402
from oracle import ref                               # noqa: E402
from test_golden import BLOCK_NAMES, load_fixture    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BASES = ("north_star_shape_N6", "random_W_nx6", "mfma_shape_nx32_N5", "constrained_nx6_nc4")
FACTOR_BLOCKS = ("ff", "fb", "Vxx", "vx")


def flat(part):
    return np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0)


def main():
    os.makedirs(os.path.join(HERE, "ref_cycle"), exist_ok=True)
    for i, name in enumerate(BASES):
        prob, mueq, _, _ = load_fixture(os.path.join(HERE, name + ".npz"))
        N = prob.horizon
        k0 = prob.stages[0]
        rng = np.random.default_rng(900 + i)
        rp = ref.Problem(prob)
        s = ref.ProximalRiccatiSolver(rp)
        assert s.backward(mueq)
        s.forward()
        cycles = N + 2
        out = {"cycles": np.int64(cycles), "mueq": np.float64(mueq)}
        for c in range(cycles):
            new = synth.generate_knot(rng, k0.nx, k0.nu, nc=k0.nc, mode="W")
            if k0.nc:
                new.D[...] = rng.uniform(-1, 1, new.D.shape)
            for b in BLOCK_NAMES:
                a = getattr(new, b)
                if a.size:
                    out[f"c{c}_knot_{b}"] = a
            rp.cycle(new)
            s.cycleAppend()
            assert s.backward(mueq)
            for nm, part in zip(("xs", "us", "vs", "lbdas"), s.forward()):
                out[f"c{c}_{nm}"] = flat(part)
            for t in range(N + 1) if c >= cycles - 2 else ():   # factors: the last two cycles (the ring has wrapped)
                f = s.datas(t)
                for b in FACTOR_BLOCKS:
                    a = getattr(f, b)
                    if a.size:
                        out[f"c{c}_s{t}_{b}"] = a
        dst = os.path.join(HERE, "ref_cycle", name + ".npz")
        np.savez_compressed(dst, **out)
        print(f"{name}: {cycles} cycles, {os.path.getsize(dst)} B")


if __name__ == "__main__":
    main()

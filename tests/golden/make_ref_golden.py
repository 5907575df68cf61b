#!/usr/bin/env python3
"""Generate tests/golden/ref/*.npz: outputs of the REFERENCE'S OWN gar code on the problems of tests/golden/*.npz.

The reference's sources (core/bunchkaufman.hpp, gar/riccati-kernel.hxx, gar/proximal-riccati.hxx,
gar/parallel-solver.hxx, gar/block-tridiagonal.hpp) are compiled UNCHANGED from /root/reference over the minimal
Eigen-API stand-in oracle/ref_shim (oracle/ref_build.sh -> oracle/_ref/libgar_ref.so; Eigen is absent from this image)
and run here; /root/reference does not exist on the GPU box, so the vectors are committed.  Every file holds, for one
problem fixture: the reference's solution, every stage's ff / fb / fth / Vxx / vx / Vxt / Vtt / vt, the Bunch-Kaufman
pivots of every stage, kkt0.ff / kkt0.fth / thGrad / thHess -- and, for the leg-parallel cases, the
ParallelRiccatiSolver's solution, per-stage factors, condensed solution and collapsed K0.
What they pin: the CPU oracle (oracle/gar_oracle.c), the kernel sources on the wave emulator, and the HIP path on
the GPU (tests/test_golden.py::test_*_reference_outputs).

Run from the repository root (in the build container):  python tests/golden/make_ref_golden.py
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref                               # noqa: E402
from test_golden import load_fixture                 # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FACTOR_BLOCKS = ("ff", "fb", "fth", "Vxx", "vx", "Vxt", "Vtt", "vt", "pivots")
PARALLEL = {"parallel_shape_nx8_N17": (2, 3, 6), "random_W_nx6": (2, 4), "north_star_shape_N6": (2,)}


def flat(part):
    return np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0)


def main():
    os.makedirs(os.path.join(HERE, "ref"), exist_ok=True)
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        name = os.path.basename(path)[:-4]
        prob, mueq, theta, _ = load_fixture(path)
        out = {"mueq": np.float64(mueq)}
        rp = ref.Problem(prob)
        s = ref.ProximalRiccatiSolver(rp)
        assert s.backward(mueq)
        for nm, part in zip(("xs", "us", "vs", "lbdas"), s.forward(theta)):
            out[nm] = flat(part)
        for t in range(prob.horizon + 1):
            f = s.datas(t)
            for b in FACTOR_BLOCKS:
                a = getattr(f, b)
                if a.size:
                    out[f"s{t}_{b}"] = a
        for nm, a in zip(("kkt0_ff", "kkt0_fth", "thGrad", "thHess"), s.initial()):
            out[nm] = a
        for legs in PARALLEL.get(name, ()):
            rpp = ref.Problem(prob)                   # ParallelRiccatiSolver mutates its problem
            par = ref.ParallelRiccatiSolver(rpp, legs)
            par.set_refinement(1e-10, 10)
            assert par.backward(mueq)
            for nm, part in zip(("xs", "us", "vs", "lbdas"), par.forward()):
                out[f"par{legs}_{nm}"] = flat(part)
            out[f"par{legs}_condensed"] = par.condensed_solution()
            for t in range(prob.horizon + 1):
                f = par.datas(t)
                for b in ("ff", "fb", "fth", "Vxx", "vx", "Vxt", "Vtt", "vt"):
                    a = getattr(f, b)
                    if a.size:
                        out[f"par{legs}_s{t}_{b}"] = a
            par.collapseFeedback()
            out[f"par{legs}_K0_collapsed"] = par.datas(0).fb
        dst = os.path.join(HERE, "ref", name + ".npz")
        np.savez_compressed(dst, **out)
        print(f"{name}: {os.path.getsize(dst)} B")


if __name__ == "__main__":
    main()

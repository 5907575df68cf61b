#!/usr/bin/env python3
"""Generate tests/golden/ref_headline/*.npz: outputs of the REFERENCE'S OWN gar code AT THE HEADLINE SIZE --
N = 256, nx = 36, nu = 12 (BASELINE.json configs[1]; the shape of tests/gar/riccati.cpp:107-139 at the bench's
horizon), generators W and F, and the reference's own benchmark shape with nc = 32 constraints per knot, mu = 1e-11
(bench/gar-riccati.cpp:19-22) -- so that parity at the configuration the metric is quoted on does not pass through
the restated oracle: the `-m gpu` tests compare gar_backward_wave<36,12> (SPD-accept on and off) and
gar_backward_wave<36,12,32> + the roll-out DIRECTLY with these files.

The reference's sources are compiled UNCHANGED from /root/reference over oracle/ref_shim (oracle/ref_build.sh ->
oracle/_ref/libgar_ref.so) and run here; /root/reference does not exist on the GPU box, so the vectors are committed.
The problems themselves are NOT stored (7.6 MB each): they are regenerated from the seed by aligator_amd.synth
(numpy's PCG64 stream), and a checksum of the packed problem in the file catches any drift of the generator.
Per file: the reference's solution xs | us | vs | lbdas, kkt0.ff, and every 32nd stage's (plus the last two)
ff / fb / Vxx / vx and Bunch-Kaufman pivots -- about 0.4 MB.

Run from the repository root (in the build container):  python tests/golden/make_ref_headline_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from aligator_amd import synth                       # noqa: E402
from oracle import ref                               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
# name -> (seed, nx, nu, nc, N, generator, mueq)
CASES = {"north_star_W_N256": (20256, 36, 12, 0, 256, "W", 1e-14),
         "north_star_F_N256": (20257, 36, 12, 0, 256, "F", 1e-14),
         "bench_nc32_N256": (20258, 36, 12, 32, 256, "W", 1e-11)}


def make_problem(name):
    seed, nx, nu, nc, N, mode, mueq = CASES[name]
    return synth.generate_lq_problem(seed, np.zeros(nx), N, nx, nu, nc=nc, mode=mode), mueq


def stages_kept(N):
    return sorted(set(range(0, N + 1, 32)) | {N - 1, N})


def problem_checksum(prob):
    """Order-dependent sums over every block of the problem: equal iff the generator reproduced the same doubles."""
    acc = []
    for t, k in enumerate(prob.stages):
        for nm in ("Q", "S", "R", "q", "r", "A", "B", "f", "C", "D", "d"):
            a = np.ravel(getattr(k, nm), order="F")
            if a.size:
                acc.append(float(np.dot(a, np.cos(np.arange(a.size) + t))))
    return np.array([np.sum(acc), np.sum(np.abs(acc)), float(np.dot(prob.g0, np.arange(prob.g0.size) + 1.0))])


def flat(part):
    return np.concatenate([np.ravel(v) for v in part]) if part else np.zeros(0)


def main():
    os.makedirs(os.path.join(HERE, "ref_headline"), exist_ok=True)
    for name in CASES:
        prob, mueq = make_problem(name)
        out = {"mueq": np.float64(mueq), "checksum": problem_checksum(prob), "case": np.array(CASES[name][:5], dtype=np.int64)}
        s = ref.ProximalRiccatiSolver(ref.Problem(prob))
        assert s.backward(mueq)
        for nm, part in zip(("xs", "us", "vs", "lbdas"), s.forward(None)):
            out[nm] = flat(part)
        for t in stages_kept(prob.horizon):
            f = s.datas(t)
            for b in ("ff", "fb", "Vxx", "vx", "pivots"):
                a = getattr(f, b)
                if a.size:
                    out[f"s{t}_{b}"] = a
        out["kkt0_ff"] = s.initial()[0]
        dst = os.path.join(HERE, "ref_headline", name + ".npz")
        np.savez_compressed(dst, **out)
        print(f"{name}: {os.path.getsize(dst)} B")


if __name__ == "__main__":
    main()

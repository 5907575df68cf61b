"""Throughput / latency of the stage-dense solver (RiccatiSolverDense, csrc/gar_dense.hpp) beside the
Riccati kernels on the same problems: north-star shape nx=36, nu=12, N=256 (the reference's
BM_stagedense, bench/gar-riccati.cpp:64-72, runs the same shape with nc=32 on the CPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver


def run(dense, batch, probs, mu, reps=5):
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], probs[0].nc0, batch=batch, dense=dense)
    s.upload([probs[b % len(probs)] for b in range(batch)])
    s.backward(mu); s.forward()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.backward_async(mu)
        s.forward_async()
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    return s.kernel_name, dt


def main():
    nx, nu, N = 36, 12, 256
    for nc, mu in ((0, 1e-14), (32, 1e-8)):
        probs = [synth.generate_lq_problem(1234 + i, np.zeros(nx), N, nx, nu, nc=nc, mode="W") for i in range(4)]
        for batch in (1, 256, 1024):
            for dense in (True, False):
                name, dt = run(dense, batch, probs, mu)
                print(f"nc={nc:2d} batch={batch:5d} {name:16s} {dt * 1e3:9.3f} ms/sweep-batch  {batch / dt:10.0f} sweeps/s",
                      flush=True)


if __name__ == "__main__":
    main()

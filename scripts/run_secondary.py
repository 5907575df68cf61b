"""Three sweeps each of the secondary shapes of bench.py (the reference's own benchmark shape nc = 32 with D = 0, the same
shape with a random D on every knot, and the Talos-walk LQ shape) at batch 1024: the process rocprofv3 --kernel-trace --stats is pointed at by scripts/gpu_r5_evidence.sh (step `secondary`)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
B = 1024
for nx, nu, nc, N, mu, coupled in ((36, 12, 32, 256, 1e-11, False), (36, 12, 32, 256, 1e-11, True), (56, 22, 0, 275, 1e-10, False)):
    probs = [synth.generate_lq_problem(100 + i, np.zeros(nx), N, nx, nu, nc=nc, mode="W") for i in range(2)]
    if coupled:   # a random D on every knot: the coupled reduced-KKT stage (gar_backward_wave_coupled)
        rng = np.random.default_rng(77)
        for p_ in probs:
            for k_ in p_.stages[:-1]:
                k_.D[...] = rng.uniform(-1.0, 1.0, k_.D.shape)
    s = BatchedRiccatiSolver([k.dims for k in probs[0].stages], nx, batch=B)
    packed = np.concatenate([s.pack(p) for p in probs])
    for b0 in range(0, B, 2):
        s.upload_packed(packed, b0, 2)
    for _ in range(3):
        s.backward_async(mu); s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    print(s.kernel_name, "ok", flush=True)
    s.close()

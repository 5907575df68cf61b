"""A/B of the pipelined sweep (gar_hip_set_pipeline) against the plain call sequence: same solver, same data,
alternating, K steps each between synchronisations; per-kernel durations from the library's events."""
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 36, 12, 256
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=batch)
st = torch.cuda.Stream()
s.set_stream(st.cuda_stream)
synth_device.fill_problems(s, seed=1, mode="W", keep=())
def run(pipe, k=steps):
    s.set_pipeline(pipe)
    for _ in range(2):
        s.backward_async(1e-14); s.forward_async()
    s.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        s.backward_async(1e-14); s.forward_async()
    s.sync(); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / k
    s._check(s._L.gar_hip_set_timing(s.handle, 1))
    kms = np.zeros(3)
    for _ in range(5):
        s.backward_async(1e-14); s.forward_async()
        o = (C.c_double * 3)()
        s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
        kms += np.array(list(o)) / 5
    s._check(s._L.gar_hip_set_timing(s.handle, 0))
    assert s.num_failed() == 0
    return t, kms
for r in range(rounds):
    for pipe in (0, 2):
        t, kms = run(pipe)
        print(json.dumps({"round": r, "pipeline": pipe, "batch": batch, "ms_per_step": t * 1e3, "sweeps_per_s": batch / t,
                          "kernel_ms(bwd,init,fwd)": [round(float(v), 3) for v in kms]}), flush=True)

"""The two roll-out kernels on the SAME factor records (same allocation, same placement), alternating: gar_forward_mfma
(252 registers, two waves per SIMD) and gar_forward_lean (LDS-DMA, 66 registers, one workgroup per CU), forward kernel
time from the library's events.  Run several times: the placement of the 26 GB of factors changes from process to
process and with it the roll-out's time (3.75 ... 4.35 ms)."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver, set_option
nx, nu, N, batch = 36, 12, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
s = BatchedRiccatiSolver(dims, nx, batch=batch)
synth_device.fill_problems(s, seed=1, mode="W", keep=())
s.backward_async(1e-14); s.forward_async(); s.sync()
s._check(s._L.gar_hip_set_timing(s.handle, 1))
s.backward_async(1e-14)   # (the getter reads the backward sweep's events too)
t = {"mfma": [], "lean": []}
for rep in range(6):
    for name in ("mfma", "lean"):
        set_option("FORWARD", None if name == "mfma" else "lean")
        s.forward_async()
        o = (C.c_double * 3)()
        s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
        if rep:
            t[name].append(o[2])
set_option("FORWARD", None)
print(json.dumps({"batch": batch, "forward_ms_mfma": round(float(np.median(t["mfma"])), 3), "forward_ms_lean": round(float(np.median(t["lean"])), 3)}), flush=True)

"""Backward-sweep kernel time (HIP events inside the library) for one or more builds of the library:
python scripts/time_backward.py <batch> lib1.so [lib2.so ...]   (A/B on one box, alternating)"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
B = int(sys.argv[1])
libs = sys.argv[2:]
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
for rep in range(2):
    for lib in libs:
        s = BatchedRiccatiSolver(dims, nx, batch=B, lib_path=os.path.join(ROOT, lib))
        synth_device.fill_problems(s, seed=1, mode=os.environ.get("GEN", "W"))
        for _ in range(2):
            s.backward_async(1e-14); s.forward_async()
        s.sync()
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        k = np.zeros(3)
        for _ in range(5):
            s.backward_async(1e-14); s.forward_async()
            o = (C.c_double * 3)()
            s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
            k += np.array(list(o))
        k /= 5
        print(f"{lib} batch {B}: bwd {k[0]:.3f} ms fwd {k[2]:.3f} ms  ({s.kernel_name})")
        s.close()
        del s

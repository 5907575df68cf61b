"""Three sweeps each of the secondary shapes of bench.py (the reference's own benchmark shape nc = 32 with D = 0, the same
shape with a random D on every knot, and the Talos-walk LQ shape) at batch 1024, every problem generated ON THE DEVICE
(aligator_amd/synth_device.py; round 5 uploaded two host problems 512 times: millions of instrumented copy dispatches under
rocprofv3 --pmc).  The process scripts/gpu_r6_evidence.sh points rocprofv3 at: --kernel-trace --stats and the FETCH_SIZE /
WRITE_SIZE / SQ counter passes.  SHAPES=0,1,2 selects; CAL=1 appends the library's streaming kernel (known byte counts:
the calibration of the counters' units, as bench.py's in-run collection does)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
B = int(os.environ.get("BATCH", "1024"))
SH = ((36, 12, 32, 256, 1e-11, False), (36, 12, 32, 256, 1e-11, True), (56, 22, 0, 275, 1e-10, False))
for i in [int(v) for v in os.environ.get("SHAPES", "0,1,2").split(",")]:
    nx, nu, nc, N, mu, coupled = SH[i]
    s = BatchedRiccatiSolver([(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)], nx, batch=B)
    synth_device.fill_problems(s, seed=100 + i, mode="W", coupled=coupled)
    for _ in range(3):
        s.backward_async(mu); s.forward_async()
    s.sync()
    assert s.num_failed() == 0
    print(s.kernel_name, "coupled" if coupled else "", "ok", flush=True)
    s.close()
if os.environ.get("CAL") == "1":   # known-size streaming kernel: 256 stages x B problems, 16 KiB read + 8 KiB written per stage
    import ctypes as C
    from aligator_amd import _lib
    L = _lib.load()
    L.gar_hip_stream_ceiling_ms.restype = C.c_double
    L.gar_hip_stream_ceiling_ms.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int]
    ms = L.gar_hip_stream_ceiling_ms(0, B, 256, 16384, 8192, 3)
    print("calibration: gar_stream_sweep reads 16384 B and writes 8192 B per stage,", B * 256, "stages per launch,", ms, "ms", flush=True)

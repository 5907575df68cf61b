"""cycleAppend as a ring (gar_hip_cycle_append on a uniform serial solver): host time of the call and time
until the stream has drained, per cycle, at the north-star shape -- nothing proportional to the problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aligator_amd import synth_device
from aligator_amd.gar import BatchedRiccatiSolver
nx, nu, N = 36, 12, 256
dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
for batch in (1, 1024):
    s = BatchedRiccatiSolver(dims, nx, batch=batch)
    synth_device.fill_problems(s, seed=3, mode="W", keep=())
    s.backward(1e-12); s.forward()
    s.cycle_append(dims[0]); s.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        s._check(s._L.gar_hip_cycle_append(s.handle, s.dims[0].ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int32))))
    t1 = time.perf_counter()
    s.sync()
    t2 = time.perf_counter()
    print(f"batch {batch}: gar_hip_cycle_append {1e6 * (t1 - t0) / 50:.1f} us per call (host), "
          f"{1e6 * (t2 - t0) / 50:.1f} us per call including the stream drain; "
          f"records copied: 0 of {N - 1} x {8 * (3684 + 3108) / 1024:.0f} KB per problem")
    s.backward(1e-12); s.forward()
    assert s.num_failed() == 0

import ctypes as C, sys
sys.path.insert(0, '/root/repo')
from aligator_amd import _lib
L = _lib.load()
L.gar_hip_stream_ceiling_ms.restype = C.c_double
L.gar_hip_stream_ceiling_ms.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int]
out = 2478 * 8
for rep in range(2):
    for name, inb in (("full 3684", 3684 * 8), ("tile-packed Q 3300", 3300 * 8), ("tri-packed Q,R 2988", 2988 * 8), ("half 1842", 1842*8)):
        ms = L.gar_hip_stream_ceiling_ms(0, 4096, 256, inb, out, 5)
        tot = 4096 * 256 * (inb + out)
        print(f"{name:22s} in={inb:6d} B  {ms:7.3f} ms  {tot/ms/1e9:7.1f} GB/s moved", flush=True)

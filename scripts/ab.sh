#!/bin/bash
# A/B two builds of the library in the same visit (same box, alternating), backward sweep ms
for rep in 1 2 3; do
  for lib in aligator_amd/libgar_hip.so ${AB_LIB:-aligator_amd/libgar_hip_norem4.so}; do
    cp $lib /tmp/lib_ab.so
    python - <<PY
import json, subprocess, sys, os
os.environ["GAR_AB"]="1"
import ctypes
sys.path.insert(0, ".")
from aligator_amd import _lib
_lib.DEFAULT_PATH = "/tmp/lib_ab.so"
import bench
sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--batch", "1024", "--no-cpu", "--no-extras"]
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print("$lib", "bwd %.3f fwd %.3f sweeps/s %.0f" % (d["kernel_ms"]["backward_sweep"], d["kernel_ms"]["forward_sweep"], d["value"]))
PY
  done
done

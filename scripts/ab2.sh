#!/bin/bash
# A/B two builds of the library in the same visit (same box, alternating): bench.py at the default batch
B=${AB_BATCH:-4096}
for rep in 1 2; do
  for lib in ${AB_LIBS:-aligator_amd/libgar_hip_v1.so aligator_amd/libgar_hip.so}; do
    GAR_AB_LIB=$PWD/$lib python - <<PY
import json, sys, os, io, contextlib
sys.path.insert(0, ".")
from aligator_amd import _lib
_lib.DEFAULT_PATH = os.environ["GAR_AB_LIB"]
import bench
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--batch", "$B", "--no-cpu", "--no-legs", "--no-extras"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print("$lib", "W: bwd %.3f fwd %.3f sweeps/s %.0f frac %.3f | F: bwd %.3f sweeps/s %.0f | err %.1e" % (
    d["kernel_ms"]["backward_sweep"], d["kernel_ms"]["forward_sweep"], d["value"], d["roofline"]["frac"],
    d["kernel_ms_F"]["backward_sweep"], d["value_F"], d["parity"]["max_rel_err_vs_oracle"]))
PY
  done
done

#!/usr/bin/env python3
"""The HIP path's MPC ring against the reference's committed cycle outputs (tests/golden/ref_cycle), without pytest
or torch (seconds on a fresh box): what tests/test_golden.py::test_hip_matches_reference_cycle_outputs asserts."""
import os
import sys
import time

t0 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden as tg  # noqa: E402

order = ["north_star_shape_N6", "random_W_nx6", "mfma_shape_nx32_N5", "constrained_nx6_nc4"]
for name in order:
    p = os.path.join(ROOT, "tests", "golden", "ref_cycle", name + ".npz")
    s = tg._hip_cycle_vs_reference(p, None)
    print(f"{name}: ok on {s.kernel_name} ({time.time() - t0:.1f} s)", flush=True)
print("all cycle fixtures ok", flush=True)

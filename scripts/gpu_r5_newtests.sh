#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_newtests; mkdir -p $O
timeout 1500 python -m pytest tests/test_integration_binding.py tests/test_multi_device.py tests/test_ref_headline.py tests/test_pipeline.py -m gpu -x -q -p no:xdist -s > $O/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "one iteration|bench/lqr.cpp|tests/lqr.cpp|seam ok|proxddp ok|MISMATCH" $O/tests.log | grep -v Warning | head -60

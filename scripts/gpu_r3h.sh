#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r3h_gpu_tests.log | tail -3
timeout 300 python scripts/time_wide_legs.py > gpurun_out/r3h_wide_legs.log 2>&1; tail -7 gpurun_out/r3h_wide_legs.log

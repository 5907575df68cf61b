#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5

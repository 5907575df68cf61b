#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|slow-path" gpurun_out/gpu_tests.log | tail -8
timeout 900 bash scripts/ab2.sh 2>&1 | grep -E "^aligator|Error|error"

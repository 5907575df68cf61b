#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "headline or both_backward or north_star or wave_family or full_size or random_large or batched_problems" > gpurun_out/gpu_tests_b.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|slow-path" gpurun_out/gpu_tests_b.log | tail -8
timeout 300 python scripts/trace_wave2.py 1024 2>&1 | tail -2
timeout 900 bash scripts/ab2.sh 2>&1 | grep -E "^aligator|Error|error" 

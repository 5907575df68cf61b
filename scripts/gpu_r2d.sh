#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_host.py -m gpu -x -q -s -k "bulk_gains or cpp_host or newton_iteration" > gpurun_out/gpu_tests_d.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|Newton" gpurun_out/gpu_tests_d.log | tail -12
./tests/cpp/_build/bench_lqr_loop 256 > gpurun_out/newton_iteration_seam.log 2>&1
./tests/cpp/_build/bench_lqr_loop 2048 >> gpurun_out/newton_iteration_seam.log 2>&1
cat gpurun_out/newton_iteration_seam.log

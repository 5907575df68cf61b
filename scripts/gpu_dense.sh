#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_solver" > gpurun_out/gpu_dense_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/gpu_dense_tests.log
timeout 600 python scripts/bench_dense.py > gpurun_out/bench_dense.log 2>&1
echo "bench rc=$?"; cat gpurun_out/bench_dense.log

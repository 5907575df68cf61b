#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "dense or generic or ldl or constrained or cstr or nc32" > gpurun_out/r3k_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r3k_gpu_tests.log | tail -3
timeout 600 python scripts/bench_gar_riccati.py > gpurun_out/r3k_gar_riccati_bench.log 2>&1; tail -17 gpurun_out/r3k_gar_riccati_bench.log

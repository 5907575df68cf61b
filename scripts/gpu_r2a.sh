#!/bin/bash
# round 2, first visit: full GPU suite, then the bench line with both generators
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|slow-path" gpurun_out/gpu_tests.log | tail -8
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_default.json

#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|Newton|north star" gpurun_out/gpu_tests.log | tail -12
./tests/cpp/_build/bench_lqr_loop 256 > gpurun_out/newton_iteration_seam.log 2>&1
./tests/cpp/_build/bench_lqr_loop 2048 >> gpurun_out/newton_iteration_seam.log 2>&1
cat gpurun_out/newton_iteration_seam.log
python scripts/cycle_append_time.py 2>&1 | tee gpurun_out/cycle_append_ring.log | tail -3
timeout 600 python bench.py --mode horizon > gpurun_out/bench_horizon_1rank.json 2> gpurun_out/bench_horizon.err; tail -c 1500 gpurun_out/bench_horizon_1rank.json; tail -3 gpurun_out/bench_horizon.err

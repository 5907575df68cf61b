#!/bin/bash
# round 3: constrained leg mode folded onto the wave-leg kernels -- parity, then the reference's own benchmark table
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "nc32 or fold" > gpurun_out/r3_fold_tests.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r3_fold_tests.log
timeout 900 python scripts/bench_gar_riccati.py > gpurun_out/r3_gar_riccati_bench.log 2>&1
echo "bench rc=$?"; cat gpurun_out/r3_gar_riccati_bench.log

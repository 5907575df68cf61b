#!/bin/bash
# round 3: blocked L D L^T building blocks -- unit tests on the GPU, the wide shape in leg mode, the stage-dense
# solver and the generic leg path of bench/gar-riccati.cpp's table
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ldl_unit.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or dense or generic or parallel" 2>&1 | tail -4
timeout 300 python scripts/time_wide_legs.py > gpurun_out/r3e_wide_legs.log 2>&1; cat gpurun_out/r3e_wide_legs.log | tail -8
timeout 600 python scripts/bench_gar_riccati.py > gpurun_out/r3e_gar_riccati_bench.log 2>&1; tail -18 gpurun_out/r3e_gar_riccati_bench.log

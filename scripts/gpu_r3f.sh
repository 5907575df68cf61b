#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ldl_unit.py -m gpu -x -q -s 2>&1 | tail -8
timeout 300 python scripts/time_wide_legs.py > gpurun_out/r3f_wide_legs.log 2>&1; tail -8 gpurun_out/r3f_wide_legs.log
timeout 600 python scripts/bench_gar_riccati.py > gpurun_out/r3f_gar_riccati_bench.log 2>&1; tail -18 gpurun_out/r3f_gar_riccati_bench.log

#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 100 python scripts/ctrace_condensed.py 2>&1 | tail -2
timeout 300 python scripts/time_wide_legs.py > gpurun_out/r3g_wide_legs.log 2>&1; tail -7 gpurun_out/r3g_wide_legs.log
timeout 600 python scripts/bench_gar_riccati.py > gpurun_out/r3g_gar_riccati_bench.log 2>&1; tail -17 gpurun_out/r3g_gar_riccati_bench.log

#!/bin/bash
# round 3, last call (a few GPU-minutes left): the wide shape in leg mode, cyclic reduction of the reduced condensed
# system against the one-workgroup chain on one box; then the GPU parity tests that run through it
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 100 python scripts/time_wide_legs.py > gpurun_out/r3r_wide_legs.log 2>&1
echo "wide legs rc=$?"; grep "^legs" gpurun_out/r3r_wide_legs.log
timeout 170 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4_talos or parallel_solver_class or batched_problems_and_legs" > gpurun_out/r3r_gpu_tests_cr.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3r_gpu_tests_cr.log

#!/bin/bash
# End-of-round evidence at the bench's default batch: gpu tests, smoke, default bench line,
# rocprofv3 kernel stats, PMC HBM traffic (calibrated), SQ MFMA-busy counters.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; B=${PBATCH:-4096}
echo "== pytest gpu =="; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/pytest_gpu.log
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
echo "== bench default =="; timeout 1200 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json | cut -c1-600
echo "== rocprof stats =="; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-legs --no-extras --single-generator > $R/gpurun_out/prof_bench.log 2>&1); head -5 gpurun_out/prof/trace_kernel_stats.csv | cut -c1-160
echo "== pmc =="; PBATCH=$B bash scripts/collect_pmc.sh 2>&1 | grep -E "hbm_bytes_per_launch|fetch_bytes|write_bytes|\"kernel\""
echo "== talos shape =="; timeout 900 python scripts/bench_talos_shape.py 256 512 1024 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/talos_shape.log | tail -9
echo "== constrained bench shape (bench/gar-riccati.cpp: nc = 32) =="; (timeout 900 python scripts/bench_constrained.py; DNONZERO=1 timeout 900 python scripts/bench_constrained.py 2>&1 | grep "D random") 2>&1 | grep -v amdgpu.ids | tee gpurun_out/constrained_bench_shape_nc32.log | cut -c1-200
echo "== seam =="; ./tests/cpp/_build/bench_lqr_loop 256 2>&1 | tee gpurun_out/newton_iteration_seam.log | tail -4
echo "== sq =="; PBATCH=$B bash scripts/collect_sq.sh 2>&1 | grep -A9 "gar_backward_wave" | head -24

#!/bin/bash
# End-of-round evidence at the bench's default batch: gpu tests, smoke, default bench line,
# rocprofv3 kernel stats, PMC HBM traffic (calibrated), SQ MFMA-busy counters.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; B=${PBATCH:-4096}
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/pytest_gpu.log
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
echo "== bench default =="; timeout 1200 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json
echo "== rocprof stats =="; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-legs --no-extras > $R/gpurun_out/prof_bench.log 2>&1); head -4 gpurun_out/prof/trace_kernel_stats.csv | cut -c1-150
echo "== pmc =="; PBATCH=$B bash scripts/collect_pmc.sh 2>&1 | grep -E "hbm_bytes_per_launch|fetch_bytes|write_bytes|\"kernel\"" 
echo "== sq =="; PBATCH=$B bash scripts/collect_sq.sh 2>&1 | grep -A9 "gar_backward_wave" | head -24

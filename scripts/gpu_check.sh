#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, benches, kernel-trace profile.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== bench (default kernel) =="; timeout 900 python bench.py --steps ${STEPS:-10} --warmup 2 --batch ${BATCH:-1024} 2>&1 | tail -1 | tee gpurun_out/bench.log
for B in ${SWEEP:-256 512 2048}; do
  echo "== bench batch $B =="; timeout 600 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_b$B.log
done
echo "== bench (generic kernel) =="; GAR_HIP_FORCE_GENERIC=1 timeout 900 python bench.py --steps 3 --warmup 1 --batch 512 --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_generic.log
echo "== rocprof =="; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 5 --warmup 1 --batch ${BATCH:-1024} --no-cpu > $R/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*.csv" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done

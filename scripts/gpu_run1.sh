#!/bin/bash
# GPU visit: re-baseline HEAD (tests, bench sweep, phase trace) + fp64 issue-rate microbenchmarks
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ubench =="; timeout 300 scripts/ubench/ubench_f64 2>&1 | tee gpurun_out/ubench_f64.log
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
for B in 256 768 1024 1536 3072; do
  echo "== bench batch $B =="; timeout 600 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_b$B.log
done
echo "== trace =="; for B in 1 256 768; do timeout 300 python scripts/trace_mfma.py $B 2>&1 | tail -6 | tee gpurun_out/trace_b$B.log; done

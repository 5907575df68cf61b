#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, a short bench, a kernel-trace profile.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" > gpurun_out/env.log
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== bench =="; timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --batch ${BATCH:-512} 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof =="; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch ${BATCH:-512} --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench default =="; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json
echo "== pmc =="; bash scripts/collect_pmc.sh 2>&1 | tail -50

#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python scripts/update_lq_rate.py 2048 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 > gpurun_out/bench6.json; cut -c1-400 gpurun_out/bench6.json

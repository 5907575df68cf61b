#!/bin/bash
set -u
export TMPDIR=/tmp
for B in 1024 2048 4096 8192; do
  timeout 900 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'batch', d['config']['batch_per_gpu'], 'sweeps/s %.0f'%d['value'], 'bwd %.3f init %.3f fwd %.3f ms'%(d['kernel_ms']['backward_sweep'], d['kernel_ms']['initial_stage'], d['kernel_ms']['forward_sweep']), 'frac %.3f'%d['roofline']['frac'])"
done

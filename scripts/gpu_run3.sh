#!/bin/bash
# 1 vs 2 waves per SIMD on the (32,12) shape, whose wave kernel fits 256 registers
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for B in 1024 2048 4096; do
  timeout 600 python bench.py --steps 5 --warmup 2 --batch $B --nx 32 --nu 12 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'batch', d['config']['batch_per_gpu'], 'sweeps/s %.0f'%d['value'], 'bwd %.3f init %.3f fwd %.3f ms'%(d['kernel_ms']['backward_sweep'], d['kernel_ms']['initial_stage'], d['kernel_ms']['forward_sweep']), 'err', d['parity']['max_rel_err_vs_oracle'])"
done

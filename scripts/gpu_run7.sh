#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/latency_legs.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tee gpurun_out/latency_legs.log

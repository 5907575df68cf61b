#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (wave) =="; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu_wave.log
for B in ${BATCHES:-1024 2048}; do
  echo "== bench wave batch $B =="; timeout 600 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'batch', d['config']['batch_per_gpu'], 'sweeps/s %.0f'%d['value'], 'bwd %.3f init %.3f fwd %.3f ms'%(d['kernel_ms']['backward_sweep'], d['kernel_ms']['initial_stage'], d['kernel_ms']['forward_sweep']), 'frac %.3f'%d['roofline']['frac'], 'err', d['parity'])" | tee gpurun_out/bench_wave_b$B.log
done
echo "== trace =="; for B in ${TRACEB:-1 1024 2048}; do timeout 300 python scripts/trace_wave.py $B 2>&1 | tail -4 | tee gpurun_out/trace_wave_b$B.log; done
if [ "${PROF:-0}" = "1" ]; then
  echo "== rocprof =="; R=$PWD; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 5 --warmup 1 --batch ${PBATCH:-1024} --no-cpu > $R/gpurun_out/prof_bench.log 2>&1)
  for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -6 $f | cut -c1-160; done
fi

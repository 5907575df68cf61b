#!/bin/bash
# HBM traffic of the sweep kernels from the rocprofv3 PMC counters, collected exactly as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (they do not fit one), counters only (no --stats / trace domains besides
# --kernel-trace), and calibrated in the same visit on known-size copies (scripts/ubench/memcal).
# Writes gpurun_out/pmc/*.csv and gpurun_out/pmc_traffic.json (copy to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; B=${PBATCH:-1024}
mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/cal_$C -o cal -- $R/scripts/ubench/memcal > $R/gpurun_out/pmc/cal_$C.log 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/bench_$C -o bench -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu --no-legs --no-extras --single-generator > $R/gpurun_out/pmc/bench_$C.log 2>&1
done
cd $R && python scripts/pmc_reduce.py gpurun_out/pmc $B | tee gpurun_out/pmc_traffic.json
# the raw per-dispatch CSVs are large (gpurun_out is capped at 64 MiB): keep the reductions only
find $R/gpurun_out/pmc $R/gpurun_out/sq -name "*.csv" -size +200k -delete 2>/dev/null

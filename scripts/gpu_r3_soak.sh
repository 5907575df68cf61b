#!/bin/bash
# randomised parity soak on the round's binary: two seeds over every family + one constrained-only run
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for S in ${SOAK_SEEDS:-301 302}; do
  SOAK_SEED=$S SOAK_SECONDS=${SOAK_SECS:-130} timeout 400 python scripts/soak.py 2>&1 | grep -E "^FAIL|^soak|^ +[0-9]+ x" | sed "s/^/seed $S: /" | tee -a gpurun_out/r3_soak_gpu.log
done
SOAK_SEED=${SOAK_CSEED:-303} SOAK_FOCUS=constrained SOAK_DENSE=0 SOAK_SECONDS=${SOAK_CSECS:-110} timeout 400 python scripts/soak.py 2>&1 | grep -E "^FAIL|^soak|^ +[0-9]+ x" | sed "s/^/constrained: /" | tee -a gpurun_out/r3_soak_gpu.log

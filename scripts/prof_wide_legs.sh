#!/bin/bash
# rocprofv3 kernel statistics of the Talos-walk shape in leg mode, one leg count (scripts/time_wide_legs.py, LEGS=16)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
L=${LEGS:-16}
mkdir -p $R/gpurun_out
cd /tmp
LEGS=$L timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_wide -o wide -- python $R/scripts/time_wide_legs.py > $R/gpurun_out/r3_prof_wide.log 2>&1
f=$(find $R/gpurun_out/r3_prof_wide -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r3_kernel_stats_wide_legs_$L.csv; head -16 "$f" | cut -c1-160; fi
find $R/gpurun_out/r3_prof_wide -type f -size +200k -delete 2>/dev/null

#!/bin/bash
# rocprofv3 kernel statistics of the Talos-walk shape in leg mode (scripts/time_wide_legs.py: serial and 2 .. 34 legs)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_wide -o wide -- python $R/scripts/time_wide_legs.py > $R/gpurun_out/r3_prof_wide.log 2>&1
f=$(find $R/gpurun_out/r3_prof_wide -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r3_kernel_stats_wide_legs.csv; head -12 "$f" | cut -c1-200; fi
find $R/gpurun_out/r3_prof_wide -type f -size +200k -delete 2>/dev/null

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/legs_prof; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/legs_prof -o legs -- python $R/scripts/prof_legs.py > $R/gpurun_out/legs_prof/run.log 2>&1
tail -1 $R/gpurun_out/legs_prof/run.log
cd $R && python - <<'PY'
import csv, glob
fs = glob.glob("gpurun_out/legs_prof/**/*kernel_stats.csv", recursive=True)
print(fs)
for r in csv.DictReader(open(fs[0])):
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
PY
find $R/gpurun_out/legs_prof -name "*.csv" -size +200k -delete 2>/dev/null

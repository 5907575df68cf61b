#!/bin/bash
# round 5, first GPU call: the pipelined sweep -- GPU parity tests, A/B against the plain sequence, kernel trace
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_pipe1
O=gpurun_out/r5_pipe1
timeout 900 python -m pytest tests/test_pipeline.py -m gpu -x -q -p no:xdist > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -5 $O/tests.log
timeout 400 python scripts/pipeline_probe.py 4096 20 3 > $O/probe_4096.log 2>&1
echo "probe rc=$?"; cat $O/probe_4096.log | tail -8
timeout 300 python scripts/pipeline_probe.py 1024 20 2 > $O/probe_1024.log 2>&1
echo "probe1024 rc=$?"; cat $O/probe_1024.log | tail -4
cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/pipeline_probe.py 4096 4 1 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/overlap.txt 2>&1 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gar_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    n = r["Kernel_Name"].split("(")[0][-40:]
    print(f'{n:42s} start {(int(r["Start_Timestamp"])-t0)/1e6:10.3f} ms  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:8.3f} ms  grid {r.get("Grid_Size","")} lds {r.get("LDS_Block_Size","")} vgpr {r.get("VGPR_Count","")} agpr {r.get("Accum_VGPR_Count","")}')
PY
cat $O/overlap.txt | tail -42

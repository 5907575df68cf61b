#!/bin/bash
# round 5: why do the half streams not overlap?  hardware-queue sharing: stream priorities / GPU_MAX_HW_QUEUES
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_pipe2
mkdir -p $O
echo "== priority streams (normal / high)"; timeout 200 python scripts/pipeline_probe.py 4096 10 1 2>&1 | grep pipeline | tee $O/prio.log
echo "== plain streams, GPU_MAX_HW_QUEUES=16"; GAR_HIP_PIPE_PRIORITY=0 GPU_MAX_HW_QUEUES=16 timeout 200 python scripts/pipeline_probe.py 4096 10 1 2>&1 | grep pipeline | tee $O/q16.log
echo "== priority streams, GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout 200 python scripts/pipeline_probe.py 4096 10 1 2>&1 | grep pipeline | tee $O/prio_q16.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/pipeline_probe.py 4096 4 1 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/overlap.txt 2>&1 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gar_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-24:]:
    n = r["Kernel_Name"].split("(")[0][-30:]
    print(f'{n:32s} q {r.get("Queue_Id","")} start {(int(r["Start_Timestamp"])-t0)/1e6:10.3f} end {(int(r["End_Timestamp"])-t0)/1e6:10.3f} dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:8.3f} ms')
PY
cat $O/overlap.txt

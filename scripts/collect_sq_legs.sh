#!/bin/bash
# MFMA-busy / VALU counters of the parallel-in-time kernels (rocprofv3 --pmc, one SQ pass, counters only)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/sq_legs; export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/sq_legs/p1 -o sq -- python $R/scripts/prof_legs.py > $R/gpurun_out/sq_legs/p1.log 2>&1
cd $R && python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_legs/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gar_" in k:
            acc[k[:64]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
    if "SQ_WAVE_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        w = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"]); m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        print(f"   => MFMA-busy / 4 / wave cycles = {m / 4 / w:.3f}")
PY
find $R/gpurun_out/sq_legs -name "*.csv" -size +200k -delete 2>/dev/null

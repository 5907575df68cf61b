#!/bin/bash
# LDS bank-conflict counters of the headline sweep kernel with V' at its unpadded pitch (the default build) and at the
# conflict-free pitch NX + 2 (make -C aligator_amd/csrc vpad): rocprofv3 --pmc (SQ block) with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; B=${PBATCH:-1024}
mkdir -p $R/gpurun_out/sqp; export TMPDIR=/tmp
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
for V in unpadded padded; do
  if [ $V = padded ]; then cp $R/aligator_amd/libgar_hip_vpad.so $R/aligator_amd/libgar_hip.so; fi
  cd /tmp
  timeout 300 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $R/gpurun_out/sqp/$V -o sq -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu --no-legs --no-extras --single-generator --pmc off > $R/gpurun_out/sqp/$V.log 2>&1
  cd $R && python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/sqp/$V/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gar_backward_wave" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("$V V' pitch:", {k: round(v) for k, v in sorted(m.items())})
if m.get("SQ_LDS_IDX_ACTIVE"):
    print("   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f ; / SQ_WAVE_CYCLES = %.3f ; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (
        m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_LDS_BANK_CONFLICT"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
done
find $R/gpurun_out/sqp -name "*.csv" -size +200k -delete 2>/dev/null

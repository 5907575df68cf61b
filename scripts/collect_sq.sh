#!/bin/bash
# MFMA-busy / VALU / wait counters of the sweep kernels (rocprofv3 --pmc, SQ block, one pass of 8)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; B=${PBATCH:-1024}
mkdir -p $R/gpurun_out/sq; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_ANY|SQ_INSTS_VALU|SQ_INSTS_LDS|SQ_INSTS_VMEM|SQ_INST_CYCLES_VMEM|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE)\b" | sort -u > $R/gpurun_out/sq/available.txt
cat $R/gpurun_out/sq/available.txt | tr '\n' ' '; echo
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/sq/p1 -o sq -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu --no-legs --no-extras --single-generator > $R/gpurun_out/sq/p1.log 2>&1
tail -2 $R/gpurun_out/sq/p1.log | cut -c1-300
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $R/gpurun_out/sq/p2 -o sq -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu --no-legs --no-extras --single-generator > $R/gpurun_out/sq/p2.log 2>&1
cd $R && python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/sq/p1", "gpurun_out/sq/p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gar_" in k:
                acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(k)
        for n, v in sorted(c.items()):
            print(f"   {n:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
# the raw per-dispatch CSVs are large (gpurun_out is capped at 64 MiB): keep the reductions only
find $R/gpurun_out/pmc $R/gpurun_out/sq -name "*.csv" -size +200k -delete 2>/dev/null

#!/bin/bash
# Round-5 evidence run (one box visit, every step under its own timeout): the GPU suite, smoke, the default bench line,
# rocprofv3 kernel stats of the bench command (both schedules: the half-batch launches carry their own kernel names),
# calibrated PMC HBM traffic of the backward AND forward sweeps, SQ counters, kernel stats of the secondary shapes.
# STEPS="tests bench prof pmc sq secondary" selects.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_evidence; mkdir -p $O; export TMPDIR=/tmp
STEPS=${STEPS:-"tests bench prof pmc sq secondary"}
cd $R
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has tests; then
  echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q -p no:xdist > $O/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
  echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error" | tee $O/smoke.log
fi
if has bench; then
  echo "== bench default =="; timeout 900 python bench.py --steps 20 --warmup 2 2> $O/bench.err | tail -1 > $O/bench_default_batch4096.json
  python - <<PY
import json; d=json.loads(open("$O/bench_default_batch4096.json").read())
print(d["value"], d["config"]["schedule"], d["roofline"]["frac"], d["kernel_ms"], {k: (v["value"] if isinstance(v, dict) else v) for k, v in d["schedules"].items()})
print("traffic", d["roofline"]["traffic"], "parity", d["parity"])
PY
fi
if has prof; then
  echo "== rocprof stats of the bench command (no extras) =="
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 2 --no-cpu --no-legs --no-extras --pmc off --single-generator > $O/prof_bench.log 2>&1)
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_batch4096.csv; head -8 $O/kernel_stats_batch4096.csv | cut -c1-170
  tail -1 $O/prof_bench.log | cut -c1-300
fi
if has pmc; then
  echo "== pmc traffic (plain schedule, batch 4096) =="
  [ -x scripts/ubench/memcal ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o scripts/ubench/memcal scripts/ubench/memcal.cpp
  mkdir -p $R/gpurun_out/pmc
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/cal_$C -o cal -- $R/scripts/ubench/memcal > $R/gpurun_out/pmc/cal_$C.log 2>&1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/bench_$C -o bench -- python $R/bench.py --steps 2 --warmup 1 --batch 4096 --no-cpu --no-legs --no-extras --single-generator --pmc off --pipeline 0 > $R/gpurun_out/pmc/bench_$C.log 2>&1)
  done
  python scripts/pmc_reduce.py gpurun_out/pmc 4096 > $O/pmc_traffic.json; grep -E "hbm_bytes_per_launch|\"kernel\"" $O/pmc_traffic.json
  find $R/gpurun_out/pmc -name "*.csv" -size +200k -delete 2>/dev/null
fi
if has sq; then
  echo "== sq counters =="
  mkdir -p $R/gpurun_out/sq
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
  (cd /tmp && timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/sq/p1 -o sq -- python $R/bench.py --steps 2 --warmup 1 --batch 4096 --no-cpu --no-legs --no-extras --single-generator --pmc off > $R/gpurun_out/sq/p1.log 2>&1)
  python - > $O/sq_counters_batch4096.log <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gar_" in k:
            acc[k[:64]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  head -40 $O/sq_counters_batch4096.log
  find $R/gpurun_out/sq -name "*.csv" -size +200k -delete 2>/dev/null
fi
if has secondary; then
  echo "== kernel stats of the secondary shapes =="
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o trace -- python $R/scripts/run_secondary.py > $O/prof_secondary.log 2>&1)
  find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_secondary_shapes.csv; head -8 $O/kernel_stats_secondary_shapes.csv | cut -c1-170
fi
find $O -name "*.csv" -size +300k -delete 2>/dev/null; rm -rf $O/prof/*/ $O/prof2/*/ 2>/dev/null

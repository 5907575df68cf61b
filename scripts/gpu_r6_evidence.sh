#!/bin/bash
# Round-6 evidence run (one box visit, every step under its own timeout; STEPS selects): the GPU suite with per-test
# output, smoke, the default bench line, rocprofv3 kernel stats of the bench command, SQ counters of the headline, and --
# new in round 6, the secondary shapes being generated on the device now -- kernel stats, FETCH_SIZE / WRITE_SIZE (separate
# passes, calibrated on known-size copies in the same visit) and SQ counters of pair<56,24>, wave<36,12,32> and the coupled
# stage (scripts/run_secondary.py).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_evidence; mkdir -p $O; export TMPDIR=/tmp
STEPS=${STEPS:-"tests bench prof sq secondary secpmc secsq multi"}
cd $R
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has tests; then
  echo "== pytest gpu (per-test output kept) =="; timeout 2400 python -m pytest tests -m gpu -q -rA -p no:xdist > $O/pytest_gpu.log 2>&1; echo "rc=$?"
  grep -E "^(PASSED|FAILED|ERROR|SKIPPED)" $O/pytest_gpu.log > $O/gpu_tests_per_test.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2 | tee -a $O/gpu_tests_per_test.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
  echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error" | tee $O/smoke.log
fi
if has bench; then
  echo "== bench default =="; timeout 1200 python bench.py --steps 20 --warmup 2 2> $O/bench.err | tail -1 > $O/bench_default_batch4096.json
  python - <<PY
import json; d=json.loads(open("$O/bench_default_batch4096.json").read())
print(d["value"], d["config"]["schedule"], d["roofline"]["frac"], d["kernel_ms"], {k: (v["value"] if isinstance(v, dict) else v) for k, v in d["schedules"].items()})
print("traffic", d["roofline"]["traffic"], "parity", d["parity"])
for k, v in d["secondary_shapes"].items(): print(k, v["kernel"], round(v["sweeps_per_s"]), v["backward_ms"], round(v["backward_frac_of_hbm_roofline"], 4), v["max_rel_err_vs_oracle"], v.get("stages_on_the_coupled_kernel_and_on_lds_bunch_kaufman"))
print("seam", {k: (v["legs"]["us_per_newton_iteration"], v["legs"]["device_ms"]) for k, v in d["seam"].items() if isinstance(v, dict) and "legs" in v})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "reference_standin")})
PY
fi
if has prof; then
  echo "== rocprof stats of the bench command (no extras) =="
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 2 --no-cpu --no-legs --no-extras --pmc off --single-generator > $O/prof_bench.log 2>&1)
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_batch4096.csv; head -8 $O/kernel_stats_batch4096.csv | cut -c1-170
  tail -1 $O/prof_bench.log | cut -c1-300
fi
if has sq; then
  echo "== sq counters, headline =="
  mkdir -p $R/gpurun_out/sq
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
  (cd /tmp && timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/sq/p1 -o sq -- python $R/bench.py --steps 2 --warmup 1 --batch 4096 --no-cpu --no-legs --no-extras --single-generator --pmc off --pipeline 0 > $R/gpurun_out/sq/p1.log 2>&1)
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
  head -24 $O/sq_counters_batch4096.log
  find $R/gpurun_out/sq -name "*.csv" -size +200k -delete 2>/dev/null
fi
if has secondary; then
  echo "== kernel stats of the secondary shapes (device-generated problems) =="
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o trace -- python $R/scripts/run_secondary.py > $O/prof_secondary.log 2>&1)
  find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_secondary_shapes.csv; head -12 $O/kernel_stats_secondary_shapes.csv | cut -c1-170
fi
if has secpmc; then
  echo "== FETCH_SIZE / WRITE_SIZE of the secondary kernels (separate passes; calibration copies in the same visit) =="
  [ -x scripts/ubench/memcal ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o scripts/ubench/memcal scripts/ubench/memcal.cpp
  P=$R/gpurun_out/secpmc; mkdir -p $P
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/cal_$C -o cal -- $R/scripts/ubench/memcal > $P/cal_$C.log 2>&1)
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/sec_$C -o sec -- python $R/scripts/run_secondary.py > $P/sec_$C.log 2>&1); echo "$C pass rc=$?"
  done
fi
if has secsq; then
  echo "== SQ counters of the secondary kernels =="
  P=$R/gpurun_out/secpmc; mkdir -p $P
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
  (cd /tmp && timeout 900 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $P/sec_sq -o sq -- python $R/scripts/run_secondary.py > $P/sec_sq.log 2>&1); echo "sq pass rc=$?"
fi
if has secpmc || has secsq; then
  python scripts/pmc_reduce_secondary.py gpurun_out/secpmc 1024 > $O/pmc_and_sq_secondary_shapes.json; python - <<PY
import json; d = json.load(open("$O/pmc_and_sq_secondary_shapes.json"))
for k, v in d["kernels"].items(): print(k, "fetch GB", round(v["fetch_bytes"] / 1e9, 2), "write GB", round(v["write_bytes"] / 1e9, 2), "traffic / algorithmic", round(v["traffic_over_algorithmic"], 3))
for k, v in d["sq"].items(): print(k[:60], "mfma busy", round(v.get("mfma_busy_over_4x_wave_cycles", 0), 3), "wait_inst_any", round(v.get("wait_inst_any_over_wave_cycles", 0), 3))
PY
  [ -s $O/pmc_and_sq_secondary_shapes.json ] && find $R/gpurun_out/secpmc -name "*.csv" -size +200k -delete 2>/dev/null
fi
if has multi; then
  echo "== scripts/first_multi_gpu.sh (one device: every rank / sub-solver on device 0) =="
  timeout 2400 bash scripts/first_multi_gpu.sh 2>&1 | grep -v amdgpu.ids | tee $O/first_multi_gpu.log | tail -40
fi
find $O -name "*.csv" -size +300k -delete 2>/dev/null; rm -rf $O/prof/*/ $O/prof2/*/ 2>/dev/null

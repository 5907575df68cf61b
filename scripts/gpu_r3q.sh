#!/bin/bash
# round 3: full GPU suite, default bench line (in-run PMC), rocprofv3 kernel stats of the bench, gar-riccati table
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3q_gpu_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3q_gpu_tests.log | tail -2
timeout 900 python bench.py > gpurun_out/r3q_bench_default.json 2> gpurun_out/r3q_bench_default.err
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3q_bench_default.json") if l.startswith("{")][0])
    r = d["roofline"]
    print("value", d["value"], "value_F", d.get("value_F"), "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"])
    print("stream", json.dumps(r["stream_ceiling"]))
    print("kernel_ms", d["kernel_ms"])
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3q_prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-legs --no-extras --single-generator --pmc off > $R/gpurun_out/r3q_prof_bench.log 2>&1
echo "rocprof rc=$?"
f=$(find $R/gpurun_out/r3q_prof -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -8 "$f" | cut -c1-220; cp "$f" $R/gpurun_out/r3q_kernel_stats_batch4096.csv; fi
find $R/gpurun_out/r3q_prof -type f -size +200k -delete 2>/dev/null
cd $R
timeout 600 python scripts/bench_gar_riccati.py > gpurun_out/r3q_gar_riccati_bench.log 2>&1
echo "gar-riccati rc=$?"; grep -v "warning\|^ *[0-9]* |\|^ *|\|In file" gpurun_out/r3q_gar_riccati_bench.log | tail -18
timeout 300 python scripts/time_wide_legs.py > gpurun_out/r3q_wide_legs.log 2>&1; tail -7 gpurun_out/r3q_wide_legs.log

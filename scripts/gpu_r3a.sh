#!/bin/bash
# round 3, first full GPU visit: the GPU suite on the refactored library, then the default bench line
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1
echo "pytest rc=$?"
tail -15 gpurun_out/r3_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r3_bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_default.json") if l.startswith("{")][0])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"])
print("traffic_source", json.dumps(r["traffic_source"])[:600])
print("stream", json.dumps(r["stream_ceiling"]))
print("secondary", json.dumps(d.get("secondary_shapes"))[:1500])
PY

#!/bin/bash
# round 3: packed Vxx in the serial family -- parity suite, then the default bench line
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3i_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r3i_gpu_tests.log | tail -3
timeout 900 python bench.py > gpurun_out/r3i_bench_default.json 2> gpurun_out/r3i_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3i_bench_default.json") if l.startswith("{")][0])
r = d["roofline"]
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 3), d["kernel_ms"])
print("frac", round(r["frac"], 4), "moved_frac", round(r["moved_frac_of_peak"], 4), "traffic/moved", r["traffic"] and round(r["traffic"] / r["moved_bytes_per_launch"], 4),
      "traffic/algorithmic", r["traffic"] and round(r["traffic"] / r["algorithmic_bytes_per_launch"], 4))
print("stream", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r["stream_ceiling"].items() if k != "note"})
print("forward GB/s", round(r["forward_GBps"]), round(r["forward_moved_GBps"]), "parity", d["parity"], "value_F", d.get("value_F"))
PY

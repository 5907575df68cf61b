#!/bin/bash
# round 3: SPD acceptance A/B on generator F, the new replay tests, headline parity
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "round3_soak or second_bunch or headline or random_large" 2>&1 | tail -4
for A in 1 0; do
  GAR_HIP_SPD_ACCEPT=$A timeout 300 python bench.py --generator F --single-generator --no-cpu --no-legs --no-extras --pmc off --steps 10 > gpurun_out/r3d_bench_F_spd$A.json 2>/dev/null
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r3d_bench_F_spd$A.json") if l.startswith("{")][0])
print("SPD_ACCEPT=$A generator F: value", round(d["value"]), "backward ms", round(d["kernel_ms"]["backward_sweep"], 3), "frac", round(d["roofline"]["frac"], 4),
      "slow", d["slow_path_stage_frac"], "pivoted", d["pivoted_stage_frac"], "parity", d["parity"])
PY
done

#!/bin/bash
# round 5, last box visit: the GPU suite on the final binary, a randomised parity soak (two seeds / focuses), one more
# default bench line.  Every step under its own timeout.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_final; mkdir -p $O; export TMPDIR=/tmp; cd $R
echo "== pytest gpu =="; timeout 900 python -m pytest tests -m gpu -q -p no:xdist > $O/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
echo "== smoke =="; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error" | tee $O/smoke.log
echo "== soak =="; SOAK_SEED=r5a SOAK_SECONDS=120 timeout 200 python scripts/soak.py 2>&1 | tail -4 | tee $O/soak_a.log
SOAK_SEED=r5b SOAK_FOCUS=constrained SOAK_SECONDS=90 timeout 170 python scripts/soak.py 2>&1 | tail -4 | tee $O/soak_b.log
echo "== bench =="; timeout 400 python bench.py --steps 20 --warmup 2 2> $O/bench.err | tail -1 > $O/bench_default_batch4096.json
python - <<PY
import json; d=json.loads(open("$O/bench_default_batch4096.json").read())
print(d["value"], d["config"]["schedule"], d["roofline"]["frac"], {k: (round(v["value"]) if isinstance(v, dict) else v) for k, v in d["schedules"].items()})
PY

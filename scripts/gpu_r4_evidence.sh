#!/bin/bash
# Round-4 evidence run (one box visit): the default bench line, rocprofv3 kernel stats of the same command, the SQ
# counters of the headline kernel at batch 4096, the seam phases, the one-process multi-device horizon line.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4_evidence; mkdir -p $O; export TMPDIR=/tmp
cd $R
echo "== bench default =="; timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default_batch4096.json; python - <<PY
import json; d=json.loads(open("$O/bench_default_batch4096.json").read())
print(d["value"], d["roofline"]["frac"], d["kernel_ms"])
PY
echo "== rocprof stats =="; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 2 --no-cpu --no-legs --no-extras --pmc off > $O/prof_bench.log 2>&1); find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_batch4096.csv; head -4 $O/kernel_stats_batch4096.csv | cut -c1-160
tail -1 $O/prof_bench.log | cut -c1-200
echo "== sq =="; PBATCH=4096 bash scripts/collect_sq.sh > $O/sq_counters_batch4096.log 2>&1; grep -A10 "gar_backward_wave" $O/sq_counters_batch4096.log | head -30
echo "== seam =="; tests/cpp/_build/seam_bench --json > $O/seam_phases.json; tests/cpp/_build/bench_lqr_loop > $O/newton_iteration_seam.log 2>&1; cat $O/newton_iteration_seam.log
echo "== horizon, one process, 2 sub-solvers on this GPU =="; timeout 600 python bench.py --mode horizon --single-process --gpus 2 --same-device 2>/dev/null | tail -1 > $O/bench_horizon_single_process_2x.json; cut -c1-600 $O/bench_horizon_single_process_2x.json
timeout 600 python bench.py --mode horizon --single-process --gpus 1 2>/dev/null | tail -1 > $O/bench_horizon_single_process_1x.json; cut -c1-300 $O/bench_horizon_single_process_1x.json
find $O -name "*.csv" -size +300k -delete 2>/dev/null; rm -rf $O/prof/*/ 2>/dev/null

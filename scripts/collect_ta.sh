#!/bin/bash
# Is the CU's vector-memory path (TA / TCP) what bounds the sweeps?  rocprofv3 --pmc, counters only (+ --kernel-trace),
# over a short run of the bench's solver (plain schedule): TA / TCP busy and stall counters beside GRBM_GUI_ACTIVE.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; B=${PBATCH:-4096}
O=$R/gpurun_out/ta; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Za-z0-9_]+|TCP_[A-Za-z0-9_]+|TD_[A-Za-z0-9_]+|TCC_[A-Z0-9_]*(BUSY|STALL|REQ|HIT|MISS)[A-Za-z0-9_]*)\b" | sort -u > $O/available.txt
wc -l $O/available.txt
pick() { for c in "$@"; do grep -qx "$c" $O/available.txt && echo -n "$c "; done; }
P1=$(pick GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum)
P2=$(pick GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum)
P3=$(pick GRBM_GUI_ACTIVE TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum)
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  [ -z "$P" ] && continue
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o ta -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu --no-legs --no-extras --single-generator --pmc off --pipeline 0 > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-200
done
cd $R && python - <<'PY' | tee gpurun_out/ta/summary.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/ta/p[0-9]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gar_" in k:
                acc[k[:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(k)
        for n, v in sorted(c.items()):
            print(f"   {n:40s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
find $O -name "*.csv" -size +200k -delete 2>/dev/null

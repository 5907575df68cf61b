#!/bin/bash
# rocprofv3 evidence for the secondary shapes (scripts/run_secondary.py): kernel stats, then HBM traffic from
# FETCH_SIZE / WRITE_SIZE in separate counter-only passes, calibrated in the same visit (as gpu_r5_evidence.sh's step "pmc").
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/pmc2; export TMPDIR=/tmp
[ -x $R/scripts/ubench/memcal ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o $R/scripts/ubench/memcal $R/scripts/ubench/memcal.cpp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc2/stats -o sec -- python $R/scripts/run_secondary.py > $R/gpurun_out/pmc2/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/cal_$C -o cal -- $R/scripts/ubench/memcal > $R/gpurun_out/pmc2/cal_$C.log 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/sec_$C -o sec -- python $R/scripts/run_secondary.py > $R/gpurun_out/pmc2/sec_$C.log 2>&1
done
cd $R && python - <<'PY' | tee gpurun_out/pmc_secondary.json
import csv, glob, json, os
root = "gpurun_out/pmc2"; GiB = 1 << 30
def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters(os.path.join(root, "cal_" + C), C).items():
        short = "b64" if "b64" in k else "b128" if "b128" in k else "seg32" if "seg32" in k else None
        if short:
            if sum(v) > 0:
                cal[(C, short)] = GiB / (sum(v) / len(v))
fetch, write = counters(os.path.join(root, "sec_FETCH_SIZE"), "FETCH_SIZE"), counters(os.path.join(root, "sec_WRITE_SIZE"), "WRITE_SIZE")
B = 1024
def alg(nx, nu, nc, N):
    knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu + nc * (nx + nu + 1)
    fac = (nu + nc + nx) * (nx + 1) + nx * nx + nx
    return 8 * (knot + fac) * N * B
res = {"batch": B, "calibration_bytes_per_count": {f"{c}:{s}": v for (c, s), v in cal.items()}, "kernels": {}}
for pat, a in (("gar_backward_wave<36, 12, 32>", alg(36, 12, 32, 256)), ("gar_backward_wave_coupled<36, 12, 32>", alg(36, 12, 32, 256)),
               ("gar_backward_pair<56, 24>", alg(56, 22, 0, 275))):
    fk = [k for k in fetch if pat in k]; wk = [k for k in write if pat in k]
    if not fk or not wk:
        continue
    # (the chain launches every kernel for every sweep: on the D = 0 problems the coupled kernel leaves at once, on the
    # D != 0 problems the decoupled one does -- the launches that did the work are the large ones)
    big = lambda v: [x for x in v if x > 0.5 * max(v)]
    f = sum(big(fetch[fk[0]])) / len(big(fetch[fk[0]])) * cal.get(("FETCH_SIZE", "seg32"), 0)
    w = sum(big(write[wk[0]])) / len(big(write[wk[0]])) * cal.get(("WRITE_SIZE", "b64"), 0)
    res["kernels"][fk[0]] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
                             "algorithmic_bytes_per_launch": a, "ratio": (f + w) / a}
print(json.dumps(res, indent=1))
PY
head -8 $R/gpurun_out/pmc2/stats/sec_kernel_stats.csv | cut -c1-170
find $R/gpurun_out/pmc2 -name "*.csv" -size +200k -delete 2>/dev/null

"""Reduce the rocprofv3 --pmc passes of scripts/gpu_r5_evidence.sh (step "pmc") to HBM bytes per launch."""
import csv, glob, json, os, sys
root, batch = sys.argv[1], int(sys.argv[2])
GiB = 1 << 30


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out


res = {"batch": batch, "unit_note": "FETCH_SIZE/WRITE_SIZE are reported in KiB-like units; the factor "
       "bytes-per-count is calibrated below on 1 GiB copies of known size in the same visit",
       "calibration": {}, "kernels": {}}
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    c = counters(os.path.join(root, "cal_" + C), C)
    for k, v in c.items():
        short = "b64" if "b64" in k else "b128" if "b128" in k else "seg32" if "seg32" in k else None
        if short:
            cal[(C, short)] = sum(v) / len(v)
            res["calibration"][f"{C}:{short}"] = {"counts_per_launch": cal[(C, short)], "bytes_moved": GiB,
                                                  "bytes_per_count": GiB / cal[(C, short)] if cal[(C, short)] else None}
# the sweep kernels read with 8 B/lane in 32 B segments (backward operands) and 16 B/lane
# (forward gains); they write 8 B/lane (gains) and 16 B/lane (Vxx)
f_r8 = res["calibration"].get("FETCH_SIZE:seg32", {}).get("bytes_per_count")
f_r16 = res["calibration"].get("FETCH_SIZE:b128", {}).get("bytes_per_count")
f_w8 = res["calibration"].get("WRITE_SIZE:b64", {}).get("bytes_per_count")
f_w16 = res["calibration"].get("WRITE_SIZE:b128", {}).get("bytes_per_count")
fetch = counters(os.path.join(root, "bench_FETCH_SIZE"), "FETCH_SIZE")
write = counters(os.path.join(root, "bench_WRITE_SIZE"), "WRITE_SIZE")
for key, pat, fr, fw in (("backward", "gar_backward", f_r8, f_w8), ("forward", "gar_forward", f_r16, f_w8),
                         ("initial", "gar_initial", f_r8, f_w8)):
    fk = [k for k in fetch if pat in k]
    wk = [k for k in write if pat in k]
    if not fk or not wk or not fr or not fw:
        continue
    fc = sum(fetch[fk[0]]) / len(fetch[fk[0]])
    wc = sum(write[wk[0]]) / len(write[wk[0]])
    res["kernels"][key] = {"kernel": fk[0][:80], "batch": batch, "fetch_counts": fc, "write_counts": wc,
                           "fetch_bytes": fc * fr, "write_bytes": wc * fw,
                           "hbm_bytes_per_launch": fc * fr + wc * fw}
print(json.dumps(res, indent=1))

"""Reduce the rocprofv3 counter passes of scripts/gpu_r6_evidence.sh (steps "secpmc", "secsq") over scripts/run_secondary.py
to per-kernel figures: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, units calibrated on the 1 GiB copies of
scripts/ubench/memcal.cpp collected in the same visit, as scripts/pmc_reduce.py does for the headline) beside each
kernel's algorithmic bytes, and the SQ counters.   usage: pmc_reduce_secondary.py ROOT BATCH"""
import csv, glob, json, os, sys
root, batch = sys.argv[1], int(sys.argv[2])
GiB = 1 << 30


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out


res = {"batch": batch, "calibration": {}, "kernels": {}}
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters(os.path.join(root, "cal_" + C), C).items():
        short = "b64" if "b64" in k else "b128" if "b128" in k else "seg32" if "seg32" in k else None
        if short and sum(v) > 1000:   # (the read-only calibration kernel writes nothing: no WRITE_SIZE:seg32)
            cal[(C, short)] = GiB / (sum(v) / len(v))
            res["calibration"][f"{C}:{short}"] = {"bytes_per_count": cal[(C, short)]}
fetch = counters(os.path.join(root, "sec_FETCH_SIZE"), "FETCH_SIZE")
write = counters(os.path.join(root, "sec_WRITE_SIZE"), "WRITE_SIZE")


def alg(nx, nu, nc, N):
    knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu + nc * (nx + nu + 1)
    fac = (nu + nc + nx) * (nx + 1) + nx * nx + nx
    return 8 * (knot + fac) * N * batch, 8 * (fac + 2 * nx + nu + nc) * N * batch


# kernel-name pattern -> (label, access widths of its loads / stores for the calibration, algorithmic bytes per launch)
bw_c, fw_c = alg(36, 12, 32, 256)
bw_t, fw_t = alg(56, 22, 0, 275)
table = [("gar_backward_pair<56, 24", "pair<56,24> backward", "seg32", "b64", bw_t),
         ("gar_forward_wide<56, 24", "forward_wide<56,24>", "b128", "b64", fw_t),
         ("gar_backward_wave<36, 12, 32", "wave<36,12,32> backward (D = 0 stages)", "seg32", "b64", bw_c),
         ("gar_backward_wave_coupled<36, 12, 32", "wave_coupled<36,12,32> backward (D != 0 stages)", "seg32", "b64", bw_c),
         ("gar_forward_mfma<36, 12, 32", "forward_mfma<36,12,32>", "b128", "b64", fw_c)]
for pat, label, rw, ww, algb in table:
    fk = [k for k in fetch if pat in k]
    wk = [k for k in write if pat in k]
    if not fk or not wk or ("FETCH_SIZE", rw) not in cal or ("WRITE_SIZE", ww) not in cal:
        continue
    # (a kernel that leaves at once -- the chain's kernels on problems they do not serve -- has launches with ~0 counts:
    # the figures are those of its busiest launches)
    fv, wv = sorted(fetch[fk[0]])[-3:], sorted(write[wk[0]])[-3:]
    fc, wc = sum(fv) / len(fv), sum(wv) / len(wv)
    fb, wb = fc * cal[("FETCH_SIZE", rw)], wc * cal[("WRITE_SIZE", ww)]
    res["kernels"][label] = {"kernel": fk[0][:90], "launches": len(fetch[fk[0]]), "fetch_counts": fc, "write_counts": wc,
                             "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                             "algorithmic_bytes_per_launch": algb, "traffic_over_algorithmic": (fb + wb) / algb}
sq = {}
for f in glob.glob(os.path.join(root, "sec_sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gar_backward" in k or "gar_forward" in k:
            sq.setdefault(k[:70], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res["sq"] = {}
for k, c in sq.items():
    m = {n: max(v) for n, v in c.items()}   # (busiest launch, see above)
    if m.get("SQ_WAVE_CYCLES"):
        m["mfma_busy_over_4x_wave_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_WAVE_CYCLES"])
        m["wait_inst_any_over_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
    res["sq"][k] = m
print(json.dumps(res, indent=1))

"""Condense the raw rocprofv3 output of scripts/profile_modes.sh into small files for profiles/.
usage: python scripts/summarize_modes.py <tag> <mode> <raw dir> <out dir>
writes <out>/<tag>_<mode>_kernel_stats.csv and <tag>_<mode>_pmc_hbm.json"""
import csv, glob, json, os, shutil, sys
tag, mode, raw, out = sys.argv[1:5]
os.makedirs(out, exist_ok=True)
plain = json.loads(open(os.path.join(raw, "plain.json")).read().strip().splitlines()[-1])
stats = sorted(glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
if stats:
    shutil.copy(stats[-1], os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, mode)))


def pmc(sub, counter):
    tot = {}
    for f in glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"].split("(")[0]
                t = tot.setdefault(k, [0.0, 0])
                t[0] += float(row["Counter_Value"]); t[1] += 1
    return tot


w, r = pmc("pmc_write", "WRITE_SIZE"), pmc("pmc_fetch", "FETCH_SIZE")
calls = plain["calls"]; cells = plain["cells_per_call"]
per = {"counts": 16, "viterbi": 1, "forward3": 8, "forward2": 8, "counts4": 16, "viterbi4": 1, "forward4b": 8, "c4b_counts": 16}[mode]
kern = {}
for k in sorted(set(w) | set(r)):
    if not (k.startswith("k_small") or "traceback" in k or "k_medium" in k):
        continue
    kern[k] = {"dispatches": w.get(k, [0, 0])[1] or r.get(k, [0, 0])[1], "write_bytes_per_call": w.get(k, [0, 0])[0] * 1024 / calls,
               "fetch_bytes_per_call_raw": r.get(k, [0, 0])[0] * 1024 / calls, "fetch_bytes_per_call_corrected": 2 * r.get(k, [0, 0])[0] * 1024 / calls}
hbm = sum(v["write_bytes_per_call"] + v["fetch_bytes_per_call_corrected"] for v in kern.values())
res = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 scripts/bench_mode.py %s 3   (separate passes for WRITE_SIZE and FETCH_SIZE)" % mode,
       "mode": mode, "workload": plain["workload"], "calls": calls, "cells_per_call": cells, "algorithmic_bytes_per_cell": per,
       "algorithmic_bytes_per_call": per * cells, "hbm_bytes_per_call": hbm, "hbm_bytes_per_cell": hbm / cells, "kernels": kern,
       "device_ms_per_call_unprofiled": plain["device_ms"],
       "note": "fetch corrected x2: on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM); the first call includes the hiprtc compile unless the disk cache is warm"}
json.dump(res, open(os.path.join(out, "%s_%s_pmc_hbm.json" % (tag, mode)), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("mode", "hbm_bytes_per_cell", "algorithmic_bytes_per_cell")}), plain["device_ms"])

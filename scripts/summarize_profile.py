"""Condense the raw rocprofv3 output of scripts/profile_round.sh into small files for profiles/.

usage: python scripts/summarize_profile.py <tag> <raw dir> <out dir>
writes <out>/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_hbm.json
"""
import csv, glob, json, os, shutil, sys

tag, raw, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)
bench = json.loads(open(os.path.join(raw, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, tag + "_bench.json"), "w"))
kernel = bench["roofline"]["kernel"]

# per-kernel stats of the profiled bench command (largest stats file = the bench process)
stats = sorted(glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
if stats:
    shutil.copy(stats[-1], os.path.join(out, tag + "_bench_kernel_stats.csv"))


def pmc_sum(sub, counter):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"] == kernel and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"]); n += 1
    return tot, n


w, nw = pmc_sum("pmc_write", "WRITE_SIZE")
r, nr = pmc_sum("pmc_fetch", "FETCH_SIZE")
cells = bench["config"]["cells_per_gpu_per_step"]
launches = nw or bench["roofline"].get("launches_per_step") or 1
res = {
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --no-cpu --no-extra --steps 1 --warmup 0  (separate passes for WRITE_SIZE and FETCH_SIZE, MI355X_MICROARCH.md 'HBM')",
    "kernel": kernel, "workload": bench["config"]["workload"],
    "WRITE_SIZE_sum_KB": w, "WRITE_SIZE_dispatches": nw, "FETCH_SIZE_sum_KB": r, "FETCH_SIZE_dispatches": nr,
    "cells_per_step": cells, "algorithmic_bytes_per_step": 8 * cells,
    "write_bytes_per_step": w * 1024, "fetch_bytes_per_step_raw": r * 1024, "fetch_bytes_per_step_corrected": 2 * r * 1024,
    "hbm_bytes_per_step": w * 1024 + 2 * r * 1024, "hbm_bytes_per_cell": (w * 1024 + 2 * r * 1024) / cells,
    "launches_per_step": launches, "hbm_bytes_per_launch": (w * 1024 + 2 * r * 1024) / launches,
    "algorithmic_bytes_per_launch": 8 * cells / launches,
    "note": "fetch corrected x2: on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM)",
}
json.dump(res, open(os.path.join(out, tag + "_pmc_hbm.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("hbm_bytes_per_cell", "launches_per_step", "hbm_bytes_per_launch")}))

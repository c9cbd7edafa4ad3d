#!/bin/bash
# Kernel trace of scripts/bench_small.py on the MI355X box: per-kernel totals (rocprofv3 --stats) and, per launch of the
# sweep kernels, the duration against the tiles it carried.
# usage: bash scripts/prof_small.sh <tag> <preset> <pairs> <inLen> <outLen>
set -u
TAG=${1:-small}; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 scripts/bench_small.py "$@" > "$OUT/run.log" 2>&1
tail -8 "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True), key=lambda p: -len(open(p).read()))[:1]:
    print(open(f).read()[:1500])
tr = sorted(glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))
if tr:
    rows = list(csv.DictReader(open(tr[0])))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", 0) or 0), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
    for k, v in by.items():
        if not k.startswith("k_small"): continue
        v.sort()
        last = v[-20:] if len(v) > 20 else v
        print(k, "launches", len(v), "vgpr", v[0][3], "lds", v[0][4])
        print("  last sweep: (grid threads, us, gap us before)")
        prev = None
        for s, e, g, _, _ in last:
            print("   %7d %8.1f %6.1f" % (g, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0)); prev = e
PY

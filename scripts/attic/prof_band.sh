#!/bin/bash
# kernel trace of the banded fills (scripts/bench_band.py): per-launch durations and the gaps between launches
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_band
rm -rf "$OUT"; mkdir -p "$OUT"
python3 scripts/bench_band.py 1024 1000 32 > "$OUT/plain.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 scripts/bench_band.py 1024 1000 32 > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive dispatches of the same kernel into sweeps (gap < 200 us)
sweeps = []; cur = None
for r in rows:
    k = r["Kernel_Name"].split("(")[0]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    if cur and cur["k"] == k and s - cur["end"] < 200000:
        cur["n"] += 1; cur["busy"] += e - s; cur["gap"] += max(0, s - cur["end"]); cur["end"] = e
    else:
        cur = {"k": k, "n": 1, "busy": e - s, "gap": 0, "start": s, "end": e}; sweeps.append(cur)
agg = collections.defaultdict(list)
for w in sweeps:
    if w["n"] >= 10: agg[(w["k"], w["n"])].append(w)
for (k, n), ws in sorted(agg.items()):
    b = sum(w["busy"] for w in ws) / len(ws) / 1e3; g = sum(w["gap"] for w in ws) / len(ws) / 1e3; span = sum(w["end"] - w["start"] for w in ws) / len(ws) / 1e3
    print("%-22s %4d launches x %3d sweeps: span %8.1f us, kernels %8.1f us, gaps %7.1f us (%.1f %%), %.1f us per launch" % (k, n, len(ws), span, b, g, 100 * g / span, b / n))
PY
tail -5 "$OUT/plain.log"

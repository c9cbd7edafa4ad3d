#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
python scripts/count_probe.py 2>&1 | tail -1
MB_MEDIUM_COUNT_FLAT=0 python scripts/count_probe.py 2>&1 | tail -1
python scripts/count_probe.py 100 487 2000 2>&1 | tail -1
python scripts/count_probe.py 16 487 2000 2>&1 | tail -1
python scripts/count_probe.py 63 487 10000 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 scripts/count_probe.py > $O/trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4b/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_medium_jit")]
g = collections.defaultdict(list)
for r in rows: g[(r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
for k, v in g.items():
    v.sort()
    n = len(v) // 4   # 4 calls (1 warm + 3)
    last = v[-n:]
    busy = sum(e - s for s, e, _ in last); span = max(e for _, e, _ in last) - min(s for s, _, _ in last)
    print("k_medium_jit lds/vgpr/wg", k, "launches/call", n, "sum ms %.2f" % (busy / 1e6), "span ms %.2f" % (span / 1e6), "avg us %.1f" % (busy / n / 1e3), "grid min/max", min(x[2] for x in last), max(x[2] for x in last))
PY

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
python scripts/count_probe.py
MB_MEDIUM_TS=128 python scripts/count_probe.py
MB_MEDIUM_TS=256 python scripts/count_probe.py
MB_MEDIUM_COUNT_G=4 python scripts/count_probe.py
MB_MEDIUM_COUNT_G=1 python scripts/count_probe.py
MB_MEDIUM_STREAMS=1 python scripts/count_probe.py
MB_MEDIUM_STREAMS=3 python scripts/count_probe.py
MB_MEDIUM_COUNTS_ROLL=0 python scripts/count_probe.py
python scripts/count_probe.py 21 487 10000
MB_MEDIUM_TS=256 python scripts/count_probe.py 21 487 10000
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 scripts/count_probe.py > $O/trace.log 2>&1
cat $(find $O/trace -name '*kernel_stats.csv' | head -1) | head -8
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3e/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_medium_jit")]
# group by LDS size = which kernel (backward fill vs count sweep)
import collections
g = collections.defaultdict(list)
for r in rows: g[(r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in g.items(): print("k_medium_jit lds/vgpr/wg", k, "launches", len(v), "total ms %.2f" % (sum(v) / 1e6), "avg us %.1f" % (sum(v) / len(v) / 1e3))
PY

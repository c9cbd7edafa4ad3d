#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" python scripts/mode_probe.py c4b 32 487 3000 cnt 2>&1 | tail -1; }
run MB_X=0
run MB_MEDIUM_COUNT_G=1
run MB_MEDIUM_COUNT_G=4
run MB_MEDIUM_COUNT_FLAT=0
MB_MEDIUM_JIT_VERBOSE=1 python scripts/mode_probe.py c4b 8 487 1000 cnt 2>&1 | grep -E "jit count|jit sum|LDS" | head
python scripts/mode_probe.py c4b 64 487 10000 fwd 2>&1 | tail -1
python scripts/mode_probe.py c4b 256 487 10000 fwd 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f/trace -o t -- python3 scripts/mode_probe.py c4b 32 487 3000 cnt > gpurun_out/r4f/trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4f/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_medium_jit")]
g = collections.defaultdict(list)
for r in rows: g[(r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
for k, v in g.items():
    v.sort(); n = len(v) // 4; last = v[-n:]
    busy = sum(e - s for s, e, _ in last); span = max(e for _, e, _ in last) - min(s for s, _, _ in last)
    print("k_medium_jit lds/vgpr/wg", k, "launches/call", n, "sum ms %.2f" % (busy / 1e6), "span ms %.2f" % (span / 1e6))
PY

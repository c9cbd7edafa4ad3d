#!/bin/bash
# SQ counters of the E-step of the 482-state machine (config 4b) at the config's STATED size, 24 pairs x 487 aa x 10000 nt (own rocprofv3 --pmc pass)
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_sq_${TAG}_c4b_counts
rm -rf "$OUT"; mkdir -p "$OUT/summary"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/pmc" -- python3 scripts/bench_mode.py c4b_counts 1 > "$OUT/run_pmc.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, os
out, tag = sys.argv[1:3]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_medium"): tot[k + " lds=" + r.get("LDS_Block_Size", "?")][r["Counter_Name"]] += float(r["Counter_Value"])
fn = os.path.join(out, "summary", "%s_c4b_counts_pmc_sq.txt" % tag)
with open(fn, "w") as g:
    g.write("tiled family, protpsw . translate . dnapsw (482 states), 24 pairs x 487 aa x 10000 nt, E-step (scripts/bench_mode.py c4b_counts 1): dispatches summed per kernel and LDS size (rocprofv3 --pmc, own pass; the Backward fill and the count sweep differ by their LDS size)\n")
    for k in sorted(tot):
        wc = tot[k].get("SQ_WAVE_CYCLES", 1.0)
        g.write("%s\n" % k)
        for c in sorted(tot[k]): g.write("  %-24s %18.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (c, tot[k][c], 100 * tot[k][c] / wc))
print(open(fn).read())
print(open(os.path.join(out, "run_pmc.log")).read()[-400:])
PY

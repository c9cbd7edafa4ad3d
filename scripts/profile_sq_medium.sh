#!/bin/bash
# tiled family, psw2dna on config 4's shape (scripts/bench_mode.py counts4 / viterbi4): kernel trace of one mode, then its SQ counters (own rocprofv3 --pmc pass).
# usage: bash scripts/profile_sq_medium.sh <counts4|viterbi4> <tag>
set -u
MODE=${1:-counts4}; TAG=${2:-r02}
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_sq_${TAG}_${MODE}
rm -rf "$OUT"; mkdir -p "$OUT/summary"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 scripts/bench_mode.py $MODE 3 > "$OUT/run.log" 2>&1
cp $(find "$OUT/trace" -name '*kernel_stats.csv' | head -1) "$OUT/summary/${TAG}_${MODE}_kernel_stats.csv"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/pmc" -- python3 scripts/bench_mode.py $MODE 3 > "$OUT/run_pmc.log" 2>&1
python3 - "$OUT" "$TAG" "$MODE" <<'PY'
import csv, glob, sys, collections, os
out, tag, mode = sys.argv[1:4]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_medium") or "traceback" in k: tot[k + " lds=" + r.get("LDS_Block_Size", "?")][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary", "%s_%s_pmc_sq.txt" % (tag, mode)), "w") as g:
    g.write("tiled family, psw2dna x 487 aa x 10000 nt (scripts/bench_mode.py %s), mode %s, dispatches summed per kernel and LDS size (rocprofv3 --pmc, own pass)\n" % (mode, mode))
    for k in sorted(tot):
        wc = tot[k].get("SQ_WAVE_CYCLES", 1.0)
        g.write("%s\n" % k)
        for c in sorted(tot[k]): g.write("  %-24s %18.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (c, tot[k][c], 100 * tot[k][c] / wc))
print(open(os.path.join(out, "summary", "%s_%s_pmc_sq.txt" % (tag, mode))).read())
print(open(os.path.join(out, "summary", "%s_%s_kernel_stats.csv" % (tag, mode))).read()[:1500])
print(open(os.path.join(out, "run.log")).read()[-600:])
PY

#!/bin/bash
# SQ counters of the small-machine family's sweeps (one rocprofv3 --pmc pass, no tracing domains): where the wave cycles of
# the compute-bound modes (traceback-byte Viterbi, rolling Forward) go.  usage: bash scripts/profile_sq.sh <tag>
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_${TAG}_sq
rm -rf "$OUT"; mkdir -p "$OUT/summary"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/pmc" -- python3 scripts/bench_small.py dnapsw 1024 1000 1000 > "$OUT/run.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, os
out, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_small") or "traceback" in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary", "%s_small_pmc_sq.txt" % tag), "w") as g:
    g.write("small-machine family, dnapsw 1024 x 1000 x 1000 (config 2), all dispatches of scripts/bench_small.py summed per kernel (rocprofv3 --pmc, own pass)\n")
    for k in sorted(tot):
        wc = tot[k].get("SQ_WAVE_CYCLES", 1.0)
        g.write("%s\n" % k)
        for c in sorted(tot[k]): g.write("  %-24s %18.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (c, tot[k][c], 100 * tot[k][c] / wc))
print(open(os.path.join(out, "summary", "%s_small_pmc_sq.txt" % tag)).read())
PY

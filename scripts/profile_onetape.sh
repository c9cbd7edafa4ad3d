#!/bin/bash
# rocprof evidence for the one-tape family (config 5 machine, 20-node profile, 64 x 2 kb): kernel stats and SQ counters.
# usage (through gpurun, from the repo root): bash scripts/profile_onetape.sh r01
set -u
TAG=${1:-r01}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_onetape_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 scripts/bench_onetape.py 20 64 ${ONETAPE_LEN:-2000} ${ONETAPE_MODES:-rmv} > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc" -- python3 scripts/bench_onetape.py 20 64 ${ONETAPE_LEN:-2000} r > "$OUT/pmc.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys, os, shutil
out, tag = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(out, "summary"), exist_ok=True)
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st: shutil.copy(st[0], os.path.join(out, "summary", "%s_onetape_kernel_stats.csv" % tag))
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
if f:
    best = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "k_wide" in r["Kernel_Name"]:
            best[r["Counter_Name"]] = max(best[r["Counter_Name"]], float(r["Counter_Value"]))   # the large dispatch of the run
    with open(os.path.join(out, "summary", "%s_onetape_pmc_sq.txt" % tag), "w") as g:
        g.write("k_wide_sweep<0> (the fused Forward || Backward launch of the cut-in-two log-likelihood), 20-node profile machine (5063 states), 64 sequences x %s nt, largest dispatch (rocprofv3 --pmc, separate pass)\n" % os.environ.get("ONETAPE_LEN", "2000"))
        wc = best.get("SQ_WAVE_CYCLES", 1.0)
        for k in sorted(best): g.write("%-24s %16.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (k, best[k], 100 * best[k] / wc))
    print(open(os.path.join(out, "summary", "%s_onetape_pmc_sq.txt" % tag)).read())
PY
grep -E "forward|viterbi fill" "$OUT/stats.log"

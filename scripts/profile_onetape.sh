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
for MODE in r v; do
  # (the generated sweeps all carry ONE kernel name: a trace per mode tells the Forward sweeps from the max sweep)
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$MODE" -- python3 scripts/bench_onetape.py 20 64 ${ONETAPE_LEN:-2000} $MODE > "$OUT/trace_$MODE.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_$MODE" -- python3 scripts/bench_onetape.py 20 64 ${ONETAPE_LEN:-2000} $MODE > "$OUT/pmc_$MODE.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc2_$MODE" -- python3 scripts/bench_onetape.py 20 64 ${ONETAPE_LEN:-2000} $MODE > "$OUT/pmc2_$MODE.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys, os, shutil
out, tag = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(out, "summary"), exist_ok=True)
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st: shutil.copy(st[0], os.path.join(out, "summary", "%s_onetape_kernel_stats.csv" % tag))
with open(os.path.join(out, "summary", "%s_onetape_pmc_sq.txt" % tag), "w") as g:
    for mode, what in (("r", "rolling log-likelihood (sequences cut in two: Forward over the prefixes || Backward over the suffixes)"), ("v", "Viterbi fill (materialised)")):
        best = {}
        for sub in ("pmc_", "pmc2_"):
            for f in glob.glob(os.path.join(out, sub + mode, "**", "*counter_collection.csv"), recursive=True):
                per = collections.defaultdict(lambda: collections.defaultdict(float))
                for r in csv.DictReader(open(f)):
                    if "k_wide" in r["Kernel_Name"]: per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                if per:      # the largest dispatch of the run
                    key = max(per, key=lambda k: max(per[k].values()))
                    best.setdefault("kernel", key[0]); best.update(per[key])
        if not best: continue
        g.write("%s, %s: 20-node profile machine (5063 states), 64 sequences x %s nt, largest dispatch (rocprofv3 --pmc, two separate passes)\n" % (best.pop("kernel"), what, os.environ.get("ONETAPE_LEN", "2000")))
        wc = best.get("SQ_WAVE_CYCLES", 1.0)
        for k in sorted(best): g.write("%-24s %16.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (k, best[k], 100 * best[k] / wc))
        g.write("\n")
print(open(os.path.join(out, "summary", "%s_onetape_pmc_sq.txt" % tag)).read())
# the figures bench.py quotes for config 5 (issue fraction of the occupied CUs): profiles/<tag>_onetape_sq.json
import json
dur = {}
for mode, key in (("v", "viterbi_fill"), ("r", "forward_cut_in_two")):
    for f in glob.glob(os.path.join(out, "trace_" + mode, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_wide_retimed" not in n and "k_wide_jit" not in n: continue
            d = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
            dur[key] = max(dur.get(key, 0.0), d)
sweeps = {}
# (round 5: k workgroups per sequence -- 64 sequences x 4 parts, or 2 halves x 2 parts: every CU holds one workgroup; MB_ONETAPE_PARTS=1: 64 / 128)
one_wg = os.environ.get("MB_ONETAPE_PARTS") == "1"
for mode, key, cus in (("v", "viterbi_fill", 64 if one_wg else 256), ("r", "forward_cut_in_two", 128 if one_wg else 256)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for sub in ("pmc_", "pmc2_"):
        for f in glob.glob(os.path.join(out, sub + mode, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_wide_retimed" in r["Kernel_Name"] or "k_wide_jit" in r["Kernel_Name"]: per[sub][r["Counter_Name"]] += float(r["Counter_Value"])
    if not per or key not in dur: continue
    # counters summed over the run's k_wide_retimed dispatches (the run calls the sweep twice: first call + timed call), time likewise
    ndisp = 2
    sweeps[key] = {"SQ_INSTS_VALU": per["pmc2_"].get("SQ_INSTS_VALU", 0.0) / ndisp, "cus": cus, "seconds": dur[key],
                   "SQ_WAIT_ANY": round(per["pmc_"].get("SQ_WAIT_ANY", 0.0) / max(per["pmc_"].get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}
json.dump({"what": "SQ_INSTS_VALU of one 64 x %s nt sweep / (SIMDs of the occupied CUs x cycles at 2.4 GHz) x 2 cycles per wave64 instruction on a SIMD-32, unweighted (fp64 at half rate: about 1.5 x); separate rocprofv3 --pmc passes, durations from the kernel trace" % os.environ.get("ONETAPE_LEN", "2000"),
           "sweeps": sweeps}, open(os.path.join(out, "summary", "%s_onetape_sq.json" % tag), "w"), indent=1)
print(json.dumps(sweeps))
PY
grep -E "forward|viterbi fill" "$OUT/stats.log"

#!/bin/bash
# Round-3 evidence in one go (through gpurun, from the repo root): the headline bench under rocprofv3 (kernel stats, HBM PMC
# passes, SQ counters of the materialised and of the matrix-free tile kernel), the two other modes of the headline machine on
# config 4's shape (kernel stats + HBM PMC + SQ), config 5 at 64 x 50 kb (kernel stats + SQ of the fused launch).
set -u
export TMPDIR=/tmp
bash scripts/profile_round.sh r03 > gpurun_out/profile_round_r03.log 2>&1; tail -2 gpurun_out/profile_round_r03.log | cut -c1-600
bash scripts/profile_modes.sh r03 counts4 viterbi4 > gpurun_out/profile_modes_r03.log 2>&1; tail -4 gpurun_out/profile_modes_r03.log | cut -c1-400
bash scripts/profile_sq_medium.sh counts4 r03 > gpurun_out/profile_sq_counts4_r03.log 2>&1; head -24 gpurun_out/profile_sq_counts4_r03.log
bash scripts/profile_sq_medium.sh viterbi4 r03 > gpurun_out/profile_sq_viterbi4_r03.log 2>&1; head -14 gpurun_out/profile_sq_viterbi4_r03.log
ONETAPE_LEN=50000 ONETAPE_MODES=rv bash scripts/profile_onetape.sh r03 > gpurun_out/profile_onetape_r03.log 2>&1; tail -14 gpurun_out/profile_onetape_r03.log
# headline kernel, SQ counters: with the matrix (bench default) and without (the same tiles, rolling mode of 64 pairs)
O=gpurun_out/prof_r03_sq; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/mat -- python3 bench.py --no-cpu --no-extra --steps 1 --warmup 0 --pairs 64 > $O/mat.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/roll -- python3 bench.py --no-cpu --no-extra --steps 1 --warmup 0 --pairs 64 --mode rolling > $O/roll.log 2>&1
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/prof_r03_sq/r03_headline_pmc_sq.txt", "w")
for tag, what in (("mat", "materialised Forward (8 B per cell stored)"), ("roll", "the same tiles WITHOUT the matrix (rolling mode, 64 pairs: halo columns + boundary records)")):
    tot = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/prof_r03_sq/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_medium_jit"): tot[r["Counter_Name"]] += float(r["Counter_Value"])
    wc = tot.get("SQ_WAVE_CYCLES", 1.0)
    out.write("k_medium_jit, psw2dna 64 x 487 aa x 10000 nt, %s; all dispatches of one step summed (rocprofv3 --pmc, own pass)\n" % what)
    for c in sorted(tot): out.write("  %-24s %18.0f  %6.1f %% of SQ_WAVE_CYCLES\n" % (c, tot[c], 100 * tot[c] / wc))
out.close()
print(open("gpurun_out/prof_r03_sq/r03_headline_pmc_sq.txt").read())
PY

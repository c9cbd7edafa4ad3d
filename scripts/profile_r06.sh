#!/bin/bash
# Round-6 evidence in one go (through gpurun, from the repo root; summaries land under gpurun_out/prof_*/summary and gpurun_out/r06/, copy them
# to profiles/):
#   1. the headline bench under rocprofv3 (kernel stats + HBM PMC passes) and the hashes of every kernel source it generated
#      (r06_kernel_sha.json: what bench.py checks its recorded constants against);
#   2. the other modes with kernel stats + HBM PMC: config 3 counts / Forward (small family), config 4 counts / Viterbi (tiled family),
#      the literal config 4 machine's Forward AND its E-step at the config's stated size (24 x 487 x 10 kb);
#   3. SQ counters: the tiled count sweep (psw2dna), the 482-state E-step at the stated size, the one-tape sweeps at 64 x 50 kb;
#   4. the generated sources of every specialised kernel for the vector-issue model (scripts/valu_model.py).
set -u
export TMPDIR=/tmp
T=r06
mkdir -p gpurun_out/$T
bash scripts/profile_round.sh $T > gpurun_out/profile_round_$T.log 2>&1; tail -2 gpurun_out/profile_round_$T.log | cut -c1-300
cp gpurun_out/prof_$T/summary_kernel_sha.json gpurun_out/$T/${T}_kernel_sha.json 2>/dev/null
cp gpurun_out/prof_$T/summary/* gpurun_out/$T/ 2>/dev/null
bash scripts/profile_modes.sh $T counts forward3 counts4 viterbi4 forward4b > gpurun_out/profile_modes_$T.log 2>&1; tail -5 gpurun_out/profile_modes_$T.log | cut -c1-300
MODE_REPS=2 bash scripts/profile_modes.sh $T c4b_counts > gpurun_out/profile_modes_c4b_$T.log 2>&1; tail -2 gpurun_out/profile_modes_c4b_$T.log | cut -c1-300
for m in counts forward3 counts4 viterbi4 forward4b c4b_counts; do cp gpurun_out/prof_${T}_$m/summary/* gpurun_out/$T/ 2>/dev/null; done
bash scripts/profile_sq_medium.sh counts4 $T > gpurun_out/profile_sq_counts4_$T.log 2>&1; head -14 gpurun_out/profile_sq_counts4_$T.log
cp gpurun_out/prof_sq_${T}_counts4/summary/*pmc_sq.txt gpurun_out/$T/ 2>/dev/null
bash scripts/profile_sq_c4b_counts.sh $T > gpurun_out/profile_sq_c4b_$T.log 2>&1; head -24 gpurun_out/profile_sq_c4b_$T.log
cp gpurun_out/prof_sq_${T}_c4b_counts/summary/* gpurun_out/$T/ 2>/dev/null
ONETAPE_LEN=50000 ONETAPE_MODES=rv bash scripts/profile_onetape.sh $T > gpurun_out/profile_onetape_$T.log 2>&1; tail -12 gpurun_out/profile_onetape_$T.log
cp gpurun_out/prof_onetape_$T/summary/* gpurun_out/$T/ 2>/dev/null
# round 6: the sweeps generated for config 5's machine against the interpreter (same process, results compared), the E-step's kernel table
# at 64 x 50 kb, what the sticky pool budget removed (scripts/stall_repro.py) and the single-pair chain (scripts/single_pair_probe.py)
timeout 900 python3 scripts/jit_probe.py 64 50000 20 roll,vit,align > gpurun_out/$T/${T}_jit_probe.txt 2>&1; tail -12 gpurun_out/$T/${T}_jit_probe.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_estep_$T -- python3 scripts/estep_probe.py 64 50000 2 > gpurun_out/$T/${T}_onetape_estep_run.txt 2>&1
cp $(find gpurun_out/prof_estep_$T -name "*kernel_stats.csv" | head -1) gpurun_out/$T/${T}_onetape_estep_kernel_stats.csv 2>/dev/null
tail -4 gpurun_out/$T/${T}_onetape_estep_run.txt
timeout 600 python3 scripts/stall_repro.py > gpurun_out/$T/${T}_stall_repro.txt 2>&1; tail -6 gpurun_out/$T/${T}_stall_repro.txt
timeout 300 python3 scripts/single_pair_probe.py dnapsw 1000 1000 200 > gpurun_out/$T/${T}_single_pair.txt 2>&1; cat gpurun_out/$T/${T}_single_pair.txt
# the generated sources the library really runs (headline shapes: 487-aa inputs select the strip width), then their ISA
J=gpurun_out/jit_$T; rm -rf $J; mkdir -p $J $J/strip
MB_JIT_CACHE=0 MB_MEDIUM_JIT_DUMP=$J/psw2dna python3 scripts/mode_probe.py psw2dna 8 487 600 fwd,roll,vit,cnt > $J/psw2dna.log 2>&1
MB_JIT_CACHE=0 MB_ROLLING_MIN_PAIRS=0 MB_MEDIUM_JIT_DUMP=$J/strip/psw2dna python3 scripts/mode_probe.py psw2dna 8 487 600 roll >> $J/psw2dna.log 2>&1
cp $J/strip/psw2dna.sum.roll.fwd.clos.hip $J/ 2>/dev/null
MB_JIT_CACHE=0 MB_MEDIUM_JIT_DUMP=$J/c4b python3 scripts/mode_probe.py c4b 8 487 600 fwd,roll,vit,cnt > $J/c4b.log 2>&1
MB_JIT_CACHE=0 MB_SMALL_JIT_DUMP=$J/protpsw python3 scripts/mode_probe.py protpsw 64 400 400 fwd,roll,vit,cnt > $J/protpsw.log 2>&1
MB_JIT_CACHE=0 MB_SMALL_JIT_DUMP=$J/dnapsw python3 scripts/mode_probe.py dnapsw 64 1000 1000 fwd,roll,vit,cnt > $J/dnapsw.log 2>&1
rm -rf $J/strip
python3 scripts/valu_model.py $J $J/${T}_valu_model.json > $J/valu_model.log 2>&1; tail -25 $J/valu_model.log
cp $J/${T}_valu_model.json gpurun_out/$T/
rm -f $J/*_full.hip
ls -la gpurun_out/$T/

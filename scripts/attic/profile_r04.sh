#!/bin/bash
# Round-4 evidence in one go (through gpurun, from the repo root; summaries land in gpurun_out/prof_*/summary, copy them to profiles/):
#   the headline bench under rocprofv3 (kernel stats + HBM PMC passes), the other modes (config 3 counts and Forward on the small
#   family, config 4 counts / Viterbi on the tiled family, the literal config 4 machine) with kernel stats + HBM PMC, SQ counters of
#   the tiled count sweep and of the one-tape sweeps at 64 x 50 kb, and the generated sources of every specialised kernel for the
#   vector-issue model (scripts/valu_model.py).
set -u
export TMPDIR=/tmp
bash scripts/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1; tail -2 gpurun_out/profile_round_r04.log | cut -c1-400
bash scripts/profile_modes.sh r04 counts forward3 counts4 viterbi4 forward4b > gpurun_out/profile_modes_r04.log 2>&1; tail -6 gpurun_out/profile_modes_r04.log | cut -c1-300
bash scripts/profile_sq_medium.sh counts4 r04 > gpurun_out/profile_sq_counts4_r04.log 2>&1; head -24 gpurun_out/profile_sq_counts4_r04.log
ONETAPE_LEN=50000 ONETAPE_MODES=rv bash scripts/profile_onetape.sh r04 > gpurun_out/profile_onetape_r04.log 2>&1; tail -24 gpurun_out/profile_onetape_r04.log
# the generated sources the library really runs (headline shapes: 487-aa inputs select the strip width), then their ISA
J=gpurun_out/jit_r04; rm -rf $J; mkdir -p $J
MB_JIT_CACHE=0 MB_MEDIUM_JIT_DUMP=$J/psw2dna python3 scripts/mode_probe.py psw2dna 8 487 600 fwd,roll,vit,cnt > $J/psw2dna.log 2>&1
MB_JIT_CACHE=0 MB_ROLLING_MIN_PAIRS=0 MB_MEDIUM_JIT_DUMP=$J/psw2dna_strip python3 scripts/mode_probe.py psw2dna 8 487 600 roll >> $J/psw2dna.log 2>&1
MB_JIT_CACHE=0 MB_MEDIUM_JIT_DUMP=$J/c4b python3 scripts/mode_probe.py c4b 8 487 600 fwd,roll,vit,cnt > $J/c4b.log 2>&1
MB_JIT_CACHE=0 MB_SMALL_JIT_DUMP=$J/protpsw python3 scripts/mode_probe.py protpsw 64 400 400 fwd,roll,vit,cnt > $J/protpsw.log 2>&1
MB_JIT_CACHE=0 MB_SMALL_JIT_DUMP=$J/dnapsw python3 scripts/mode_probe.py dnapsw 64 1000 1000 fwd,roll,vit,cnt > $J/dnapsw.log 2>&1
python3 scripts/valu_model.py $J $J/r04_valu_model.json > $J/valu_model.log 2>&1; tail -30 $J/valu_model.log
rm -f $J/*_full.hip

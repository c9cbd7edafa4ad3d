#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" python scripts/mode_probe.py c4b 32 487 3000 fwd,roll 2>&1 | tail -1; }
run MB_X=0
run MB_MEDIUM_CLOSURE_STAGES=3
run MB_MEDIUM_CLOSURE_STAGES=4
run MB_MEDIUM_CLOSURE_STAGES=6
run MB_MEDIUM_CLOSURE_STAGES=8
run MB_MEDIUM_G=4
run MB_MEDIUM_G=1
run MB_JIT_STAGE_MAXLOADS=16
run MB_JIT_STAGE_MAXLOADS=64
run MB_JIT_MAXCANDS=8
run MB_JIT_MAXCANDS=16
run MB_MEDIUM_TS=128
python scripts/mode_probe.py c4b 64 487 10000 fwd,vit 2>&1 | tail -1
python scripts/mode_probe.py c4b 24 487 10000 cnt 2>&1 | tail -1

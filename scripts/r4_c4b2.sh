#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" python scripts/mode_probe.py c4b 32 487 3000 fwd,roll 2>&1 | tail -1; }
run MB_MEDIUM_CLOSURE_STAGES=0
run MB_MEDIUM_CLOSURE_STAGES=5
run MB_MEDIUM_CLOSURE_STAGES=7
run MB_MEDIUM_CLOSURE_STAGES=10
run MB_MEDIUM_CLOSURE_STAGES=12
run MB_MEDIUM_SPLIT_DEGREE=8
run MB_MEDIUM_SPLIT_DEGREE=24
run MB_MEDIUM_MAXWAVES=6
MB_MEDIUM_JIT_VERBOSE=1 python scripts/mode_probe.py c4b 8 487 1000 roll 2>&1 | grep -E "stage cuts|closure stages|register budget" | head -30

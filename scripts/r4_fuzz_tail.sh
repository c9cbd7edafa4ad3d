#!/bin/bash
# the knob-fuzz lines added after the 126-slot usage pass was found (scripts/fuzz_knobs.sh holds them too)
f() { echo "== $*"; env "$@" python scripts/fuzz_gpu.py ${CASES:-60} ${SEED} 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|RuntimeWarning\|ok = " | cut -c1-240 | tail -4; }
SEED=68000 f MB_JIT_FLAT_CHUNK=4 MB_MEDIUM_G=16
SEED=69000 f MB_JIT_FLAT_CHUNK=0 MB_MEDIUM_G=16
SEED=70000 f MB_MEDIUM_COUNT_G=16
SEED=71000 f MB_MEDIUM_COUNT_G=32 MB_JIT_FLAT_CHUNK=7
SEED=72000 f MB_MEDIUM_G=32
SEED=73000 f MB_MEDIUM_G=4 MB_JIT_REGBUDGET=0

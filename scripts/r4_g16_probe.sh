#!/bin/bash
# round 4: the flat count program at 16 columns per wavefront (fuzz case 46000/19): batched usage pass against the unbatched one under
# different spilling strategies of the compiler
run() { echo "== $*"; env "$@" MB_MEDIUM_G=16 MB_MEDIUM_JIT_VERBOSE=1 python scripts/fuzz_one_counts.py 46000 19 2>&1 | grep -E "family 0|spill" | cut -c1-200 | tail -4; }
run A=1
run MB_JIT_FLAT_CHUNK=0
run MB_JIT_FLAT_CHUNK=0 "MB_JIT_EXTRA_OPTS=-mllvm -amdgpu-spill-vgpr-to-agpr=0"
run MB_JIT_FLAT_CHUNK=0 "MB_JIT_EXTRA_OPTS=-O1"
run MB_JIT_FLAT_CHUNK=48

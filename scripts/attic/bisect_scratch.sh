#!/bin/bash
# Bisect of round 2's "the 469-spill build gives wrong Forward values" (scripts/fuzz_env_gpu.py seed 5229); result in
# profiles/r03_scratch_bisect.txt.  Prepare in the container (hipcc cross-compiles; the built trees travel with gpurun):
#   mkdir _bisect; for x in A C D; do git worktree add -f _bisect/$x f36f55b^; done; git worktree add -f _bisect/B f36f55b
#   cp _bisect/B/scripts/fuzz_env_gpu.py _bisect/{A,C,D}/scripts/
#   C: the one-line LDS-block fix of f36f55b in mb_small.cpp (wave_doubles);  D: `git show f36f55b -- machineboss_amd/csrc/mb_api.hip | git apply`
#   (cd _bisect/$x && python -m machineboss_amd.build) for each; then on the GPU box: bash scripts/bisect_scratch.sh
run() { local d=$1; shift; local label=$1; shift
  echo "== $label"
  ( cd $d && env MB_JIT_CACHE=0 MB_SMALL_JIT_VERBOSE=1 "$@" timeout 300 python scripts/fuzz_env_gpu.py 1 5229 2>&1 | grep -E "MISMATCH|cases,|mode 3" | sed 's/^/   /' ) }
for rep in 1 2; do run _bisect/A "A = f36f55b^ (before the three fixes), default (3 waves/SIMD for the enveloped count sweep), rep $rep"; done
run _bisect/A "A, MB_SMALL_MINWAVES=2" MB_SMALL_MINWAVES=2
run _bisect/A "A, MB_SMALL_MINWAVES=1" MB_SMALL_MINWAVES=1
for rep in 1 2; do run _bisect/B "B = f36f55b (LDS block + halo widths fixed), scratch allowed, 3 waves/SIMD, rep $rep" MB_SMALL_ALLOW_SCRATCH=1; done
run _bisect/B "B, scratch allowed, MB_SMALL_MINWAVES=4" MB_SMALL_ALLOW_SCRATCH=1 MB_SMALL_MINWAVES=4
run _bisect/B "B, default (no scratch)"
for rep in 1 2 3; do run _bisect/C "C = f36f55b^ + ONLY the LDS block fix, 3 waves/SIMD (469 spills), rep $rep"; done
for rep in 1 2 3; do run _bisect/D "D = f36f55b^ + ONLY the halo / boundary width fix, 3 waves/SIMD (469 spills), rep $rep"; done

#!/bin/bash
# Profile the default bench.py command on the MI355X box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats      -> per-kernel time
#   2. rocprofv3 --pmc WRITE_SIZE            -> HBM write bytes   (separate passes, no tracing domains with --pmc)
#   3. rocprofv3 --pmc FETCH_SIZE            -> HBM fetch bytes (x2 on gfx950, MI355X_MICROARCH.md "HBM")
# then scripts/summarize_profile.py turns the raw csv files under gpurun_out/ into profiles/<tag>_*.
# usage: bash scripts/profile_round.sh <tag>     e.g. r01
set -u
TAG=${1:-r01}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --write-kernel-sha "$OUT/summary_kernel_sha.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu --no-extra > "$OUT/stats.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --no-cpu --no-extra --steps 1 --warmup 0 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu --no-extra --steps 1 --warmup 0 > "$OUT/pmc_fetch.log" 2>&1
python3 scripts/summarize_profile.py "$TAG" "$OUT" "$OUT/summary" || true
cat "$OUT/bench.json"

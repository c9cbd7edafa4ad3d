#!/bin/bash
# rocprofv3 evidence for the non-headline modes (run through gpurun from the repo root):
#   kernel trace + stats, then --pmc WRITE_SIZE and --pmc FETCH_SIZE in passes of their own (no tracing domain with --pmc),
#   each on `python3 scripts/bench_mode.py <mode> 3`; scripts/summarize_modes.py condenses them into profiles/<tag>_<mode>_*.
# usage: bash scripts/profile_modes.sh <tag> <mode> [<mode> ...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
REPS=${MODE_REPS:-3}
for MODE in "$@"; do
  OUT=$(pwd)/gpurun_out/prof_${TAG}_$MODE
  rm -rf "$OUT"; mkdir -p "$OUT"
  python3 scripts/bench_mode.py $MODE $REPS > "$OUT/plain.json" 2> "$OUT/plain.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 scripts/bench_mode.py $MODE $REPS > "$OUT/stats.json" 2> "$OUT/stats.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 scripts/bench_mode.py $MODE $REPS > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 scripts/bench_mode.py $MODE $REPS > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
  python3 scripts/summarize_modes.py "$TAG" "$MODE" "$OUT" "$OUT/summary" || true
done

#!/bin/bash
# round 4: the count sweep of the 482-state machine (config 4b), its geometry as the library chose it and a few alternatives
MB_MEDIUM_JIT_VERBOSE=1 MB_TIMING=1 python scripts/mode_probe.py c4b 32 487 3000 cnt 2>&1 | grep -v "closure stages\|program cost" | tail -25
for kv in MB_MEDIUM_COUNT_G=2 MB_JIT_PLACE=0 MB_JIT_REGBUDGET=0 MB_MEDIUM_COUNT_FLAT=0; do
  env $kv python scripts/mode_probe.py c4b 32 487 3000 cnt 2>&1 | tail -1
done

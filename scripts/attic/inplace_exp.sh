#!/bin/bash
for kv in "MB_MEDIUM_INPLACE_RING=0" "MB_MEDIUM_INPLACE_RING=1 MB_MEDIUM_COMPACT_MAXWAVES=8"; do
  env $kv MB_ROLLING_MIN_PAIRS=100000 python scripts/mode_probe.py psw2dna 64 487 2000 roll 2>&1 | tail -1 | cut -c1-200
  env $kv python scripts/mode_probe.py psw2dna 256 487 10000 roll 2>&1 | tail -1 | cut -c1-200
  env $kv MB_MEDIUM_JIT_VERBOSE=1 python scripts/mode_probe.py c4b 64 487 3000 roll 2>&1 | grep "in-place\|c4b 64" | cut -c1-200
done

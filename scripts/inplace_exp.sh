#!/bin/bash
for w in 9 10 11; do
  MB_MEDIUM_COMPACT_MAXWAVES=$w MB_ROLLING_MIN_PAIRS=100000 python scripts/mode_probe.py psw2dna 64 487 2000 roll 2>&1 | tail -1 | cut -c1-200
  MB_MEDIUM_COMPACT_MAXWAVES=$w python scripts/mode_probe.py psw2dna 256 487 10000 roll 2>&1 | tail -1 | cut -c1-200
done

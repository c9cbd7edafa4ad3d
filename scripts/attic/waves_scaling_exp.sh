#!/bin/bash
# does a tile kernel's rate scale with resident wavefronts?  psw2dna, 64 x 487 x 2000: rolling tiles (log-likelihood), traceback-byte Viterbi, E-step
for w in 2 4 6 8; do
  MB_MEDIUM_MAXWAVES=$w MB_ROLLING_MIN_PAIRS=100000 python scripts/mode_probe.py psw2dna 64 487 2000 roll,vit 2>&1 | tail -1 | cut -c1-230
done
for w in 2 4 6 8; do
  MB_MEDIUM_COUNT_MAXWAVES=$w python scripts/mode_probe.py psw2dna 64 487 2000 cnt 2>&1 | tail -1 | cut -c1-160
done

#!/bin/bash
# round 4: count sweep of the 482-state machine with more than 8 wavefronts (= columns, G = 1) per workgroup: the records that no
# longer fit the LDS beside the extra columns are read through buffer loads
for w in 8 9 10 12; do
  MB_MEDIUM_COUNT_MAXWAVES=$w MB_MEDIUM_JIT_VERBOSE=1 python scripts/mode_probe.py c4b 32 487 3000 cnt 2>&1 | grep -E "jit count|^c4b" | tail -2
done
MB_MEDIUM_COUNT_MAXWAVES=16 MB_MEDIUM_COUNT_G=2 python scripts/mode_probe.py c4b 32 487 3000 cnt 2>&1 | tail -1
MB_MEDIUM_COUNT_MAXWAVES=12 python scripts/mode_probe.py psw2dna 64 487 2000 cnt 2>&1 | tail -1
python scripts/mode_probe.py psw2dna 64 487 2000 cnt 2>&1 | tail -1

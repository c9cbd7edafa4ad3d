#!/bin/bash
# C4b count sweep: two columns per wavefront with the lighter LDS footprint
for kv in "MB_MEDIUM_COUNT_G=2" "MB_MEDIUM_COUNT_G=2 MB_MEDIUM_COUNT_MAXWAVES=4" "MB_MEDIUM_COUNT_G=2 MB_MEDIUM_COUNT_MAXWAVES=5" "MB_MEDIUM_COUNT_G=2 MB_MEDIUM_COUNT_MAXWAVES=4 MB_MEDIUM_SYNC_COST=3" "MB_MEDIUM_COUNT_G=4 MB_MEDIUM_COUNT_MAXWAVES=4" "MB_MEDIUM_COUNT_MAXWAVES=4 MB_MEDIUM_SYNC_COST=3"; do
  rm -f /tmp/xd.*; env $kv MB_MEDIUM_JIT_VERBOSE=1 MB_MEDIUM_JIT_DUMP=/tmp/xd python scripts/mode_probe.py c4b 16 487 3000 cnt > /tmp/o.txt 2>&1
  echo "$(tail -1 /tmp/o.txt | cut -c1-90) | $(grep 'jit count' /tmp/o.txt | tail -1) | $(grep -E '^#define (JC|JG|JLDSRECS|JNACC) ' /tmp/xd.cnt.tiles.fwd.clos.hip | tr '\n' ' ') $kv"
done

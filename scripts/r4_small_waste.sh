#!/bin/bash
# round 4: what the skew (64 steps per strip sweep) and the padding columns of the last strip cost the small family on config 3's
# shape -- the same batch at lengths that fill the last strip (383), that waste most of it (385), and config 3's own (400)
for shape in "383 400" "400 400" "385 400" "447 400" "400 383" "400 1000" "383 1000"; do
  set -- $shape
  echo "== protpsw 1024 x $1 x $2"; python scripts/bench_small.py protpsw 1024 $1 $2 2>&1 | grep -E "materialised|counts|rolling"
done
echo "== TS=128, 400 x 400"; MB_SMALL_TS=128 python scripts/bench_small.py protpsw 1024 400 400 2>&1 | grep -E "materialised|counts"
echo "== 4096 pairs, 400 x 400"; python scripts/bench_small.py protpsw 4096 400 400 2>&1 | grep -E "materialised|counts"

#!/bin/bash
# round 4: one-tape --align through traceback codes (two-butterfly reduction, 32-bit code address, walker with fast words) against the
# fp64 route, at config 5's size and at small ones (which route should be the default?)
for i in 1 2; do MB_ONETAPE_TB=1 python scripts/bench_onetape.py 20 64 50000 vp 2>&1 | grep -E "with paths|device|bit-exact"; done
MB_ONETAPE_TB=1 MB_ONETAPE_TB_FAST=0 python scripts/bench_onetape.py 20 64 50000 vp 2>&1 | grep -E "with paths|device"
for shape in "64 2000" "256 4000" "8 500"; do
  for tb in 1 0; do echo "== $shape TB=$tb"; MB_ONETAPE_TB=$tb python scripts/bench_onetape.py 20 $shape vp 2>&1 | grep -E "with paths|device"; done
done
for tb in 1 0; do echo "== whole fn3 16 x 3000 TB=$tb"; MB_ONETAPE_TB=$tb python scripts/bench_onetape.py 86 16 3000 vp 2>&1 | grep -E "with paths|device"; done
python -m pytest tests/test_gpu_parity.py -x -q -k "one_tape or onetape or config5 or retimed" 2>&1 | tail -3

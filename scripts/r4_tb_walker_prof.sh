#!/bin/bash
# kernel times of the one-tape traceback-code route at 64 x 50 kb, walker with and without the fast words
export TMPDIR=/tmp
for fast in 1 0; do
  rm -rf /tmp/walkprof; MB_ONETAPE_TB=1 MB_ONETAPE_TB_FAST=$fast rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/walkprof -o t -- python3 scripts/bench_onetape.py 20 64 50000 vp > /tmp/walk.log 2>&1
  echo "== fast=$fast"; f=$(find /tmp/walkprof -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-220
done

#!/bin/bash
# kernel times of the one-tape traceback-code route at 64 x 50 kb (the walker on the scalar unit), then the one-tape parity tests
export TMPDIR=/tmp
for fast in 1 0; do
  rm -rf /tmp/walkprof; MB_ONETAPE_TB_FAST=$fast rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/walkprof -o t -- python3 scripts/bench_onetape.py 20 64 50000 vp > /tmp/walk.log 2>&1
  echo "== fast=$fast"; f=$(find /tmp/walkprof -name '*kernel_stats.csv' | head -1); grep -E "traceback_codes|true>" "$f" | cut -c1-60,150-260; grep -E "with paths|device|bit-exact" /tmp/walk.log
done
python scripts/bench_onetape.py 20 64 2000 vp 2>&1 | grep -E "with paths|device"
python scripts/bench_onetape.py 86 16 3000 vp 2>&1 | grep -E "with paths|device"
python -m pytest tests/test_gpu_parity.py -x -q -k "one_tape or onetape or config5 or retimed or wide" 2>&1 | tail -3

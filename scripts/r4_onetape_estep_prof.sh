#!/bin/bash
# kernels of the one-tape E-step (config 5 machine): 64 x 2 kb and 64 x 10 kb
export TMPDIR=/tmp
for len in 2000 10000; do
  rm -rf /tmp/esprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/esprof -o t -- python3 scripts/bench_onetape.py 20 64 $len c > /tmp/es.log 2>&1
  echo "== 64 x $len"; f=$(find /tmp/esprof -name '*kernel_stats.csv' | head -1); head -9 "$f" | awk -F'","' '{print substr($1,1,70), $2, $3, $4}' ; grep "fwd+bwd" /tmp/es.log | cut -c1-200
done

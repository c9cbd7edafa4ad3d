#!/bin/bash
# rocprofv3 kernel trace of the one-tape family on the 20-node config-5 machine, 64 x 2 kb: log-likelihood (sequences cut in
# two, both halves in one launch + k_onetape_join) and the count call (fused Forward + Backward fill, LDS-staged count kernel)
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=$(pwd)/gpurun_out/prof_${TAG}_onetape
rm -rf "$OUT"; mkdir -p "$OUT/summary"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 scripts/bench_onetape.py 20 64 2000 rc > "$OUT/run.log" 2>&1
cp "$OUT/trace/t_kernel_stats.csv" "$OUT/summary/${TAG}_onetape_kernel_stats.csv"
grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" "$OUT/run.log" | grep "Gcells\|composed" > "$OUT/summary/${TAG}_onetape_run.txt"
cat "$OUT/summary/${TAG}_onetape_run.txt"; head -8 "$OUT/summary/${TAG}_onetape_kernel_stats.csv" | cut -c1-200

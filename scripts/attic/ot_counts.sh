cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MB_ONETAPE_COUNTS=0 python scripts/bench_onetape.py 20 64 2000 c 2>&1 | grep "fwd+bwd"
python scripts/bench_onetape.py 20 64 2000 c 2>&1 | grep "fwd+bwd"
rocprofv3 --kernel-trace --stats -d gpurun_out/ot_prof -o ot -- python3 scripts/bench_onetape.py 20 64 2000 c > gpurun_out/ot_prof.log 2>&1
python3 - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/ot_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.reader(open(f)))[:8]: print(r[:5])
PY
python scripts/bench_onetape.py 86 16 500 c 2>&1 | grep "fwd+bwd\|composed"
MB_ONETAPE_COUNTS=0 python scripts/bench_onetape.py 86 16 500 c 2>&1 | grep "fwd+bwd"

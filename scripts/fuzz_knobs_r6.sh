# round 6: the generated one-tape sweeps (levels, log-sum-exp form, interpreter), the small family's persistent strips (hand-over block,
# strips per workgroup, tiles in one grid, launch by launch), the third-pass usage kernel, tile heights, a small pool budget;
# same harness as fuzz_knobs.sh
f() { echo "== $*"; env "$@" python scripts/fuzz_gpu.py ${CASES:-60} ${SEED} 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|RuntimeWarning\|ok = " | cut -c1-240 | tail -4; }
SEED=91000 f MB_WIDE_JIT=0 MB_WIDE_MIN_STATES=1
SEED=92000 f MB_WIDE_JIT_LEVEL=1 MB_WIDE_MIN_STATES=1
SEED=93000 f MB_WIDE_JIT_TWOPASS=0 MB_WIDE_MIN_STATES=1
SEED=94000 f MB_WIDE_MIN_STATES=1 MB_ONETAPE_PARTS_MIN_LEN=0 MB_ONETAPE_TRACEBACK_MIN_TRANS=0
SEED=95000 f MB_SMALL_ONE_LAUNCH=2 MB_SMALL_JSUB=2
SEED=96000 f MB_SMALL_ONE_LAUNCH=2 MB_SMALL_JSUB=64 MB_SMALL_STRIP_PER_WG=0
SEED=97000 f MB_SMALL_ONE_LAUNCH=1
SEED=98000 f MB_SMALL_ONE_LAUNCH=0
SEED=99000 f MB_MEDIUM_COUNT_PASSES=3
SEED=100000 f MB_MEDIUM_TS_LONG_MIN_PAIRS=0 MB_MEDIUM_TS=32
SEED=101000 f MB_MEM_FRACTION=0.002
SEED=102000 f MB_ONETAPE_COUNT_FP64=1 MB_WIDE_MIN_STATES=1 MB_ONETAPE_PARTS_MIN_LEN=0

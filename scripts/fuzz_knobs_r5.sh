# round 5: the count programs' new forms (fused emit usage, two accumulator tables) and the fp64 correction term of the one-tape E-step,
# under the knobs that move their placement / geometry; same harness as fuzz_knobs.sh
f() { echo "== $*"; env "$@" python scripts/fuzz_gpu.py ${CASES:-60} ${SEED} 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|RuntimeWarning\|ok = " | cut -c1-240 | tail -4; }
SEED=80000 f MB_X=0
SEED=81000 f MB_MEDIUM_COUNT_FUSE=0
SEED=82000 f MB_MEDIUM_COUNT_COMPACT=0
SEED=83000 f MB_JIT_REGBUDGET=0
SEED=84000 f MB_JIT_REGBUDGET=30 MB_MEDIUM_COUNT_G=1
SEED=85000 f MB_MEDIUM_COUNT_G=8 MB_MEDIUM_SPLIT_DEGREE=4
SEED=86000 f MB_MEDIUM_COUNT_G=32 MB_DETERMINISTIC=1
SEED=87000 f MB_ONETAPE_COUNT_FP64=1 MB_WIDE_MIN_STATES=1
SEED=88000 f MB_ONETAPE_COUNT_FP64=1 MB_WIDE_GLOBAL_VECTORS=1
SEED=89000 f MB_MEDIUM_COUNTS_ROLL=0
SEED=90000 f MB_MEDIUM_COUNT_MAXWAVES=3 MB_MEDIUM_CLOSURE_STAGES=3

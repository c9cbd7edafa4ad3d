"""Randomised parity sweep on the GPU: random advancing machines (two-tape and one-tape, duplicate edges, -inf weights,
uneven silent levels) and ragged random batches through the automatically chosen kernel family against the generic family
(fills, rolling log-likelihood, Viterbi paths, counts), and against the C oracle for the first pair of every case.
usage: python scripts/fuzz_gpu.py [cases] [seed0]"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_machine, random_seq

def close(a, b, rel, abs_):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ok = (np.isneginf(a) & np.isneginf(b)) | (np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= abs_ + rel * np.abs(b)))
    return bool(np.all(ok))

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for c in range(cases):
    rng = np.random.RandomState(seed0 + c)
    oneTape = c % 3 == 2
    S = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 16, 17, 33, 64, 100, 257, 300, 700])) if not oneTape else int(rng.choice([5, 40, 260, 300, 900]))
    nIn = 0 if oneTape else int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
    if oneTape and c % 6 == 5: nIn, nOut = nOut, 0          # a recogniser: the one tape is the input
    em = random_machine(S, nIn, nOut, seed0 + c, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=bool(c % 5 == 0))
    if oneTape: os.environ["MB_WIDE_MIN_STATES"] = "1" if c % 2 else "256"
    dm = capi.DeviceMachine(em); om = oracle.OracleMachine(em)
    n = int(rng.randint(1, 6))
    scale = 6 if (c % 7 == 3 and S <= 100) or (c % 2 == 1 and S <= 16) else 1   # longer lattices (several tiles / strips) on the smaller machines
    lo = 64 if oneTape and c % 4 == 2 else 0            # one-tape batches of sequences >= 64 symbols are cut in two (k_onetape_join)
    pairs = [(random_seq(rng, int(rng.randint(lo, lo + 40 * scale)) if nIn else 0, nIn), random_seq(rng, int(rng.randint(lo, lo + 60 * scale)) if nOut else 0, nOut)) for _ in range(n)]
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    cut = oneTape and min(max(len(x), len(y)) for x, y in pairs) >= 64      # the library cuts such batches in two
    out = {}
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
        capi.set_kernel(fam)
        try:
            x, y = pairs[0]
            out[fam] = dict(F=dm.fill(capi.MB_FORWARD, x, y), B=dm.fill(capi.MB_BACKWARD, x, y), V=dm.fill(capi.MB_VITERBI, x, y),
                            ll=b.forward(capi.MB_ROLLING), llm=b.forward(capi.MB_MATERIALISE), vit=b.viterbi(), cnt=b.counts(), kern=capi.last_kernel_name())
        finally:
            capi.set_kernel(capi.KERNEL_AUTO)
    a, g = out[capi.KERNEL_AUTO], out[capi.KERNEL_GENERIC]
    x, y = pairs[0]
    Vo = om.viterbi(x, y)
    checks = dict(V=np.array_equal(a["V"], g["V"]) and np.array_equal(a["V"], Vo),
                  F=close(a["F"], g["F"], 2e-6, 2e-5) and close(a["F"], om.forward(x, y, oracle.SUM_EXACT), 2e-6, 2e-5),
                  B=close(a["B"], g["B"], 2e-6, 2e-5), ll=close(a["ll"], g["ll"], 2e-6, 2e-5) and close(a["ll"], a["llm"], 2e-6 if cut else 1e-9, 2e-5 if cut else 1e-12),   # (sequences cut in two: another summation order)
                  vit=np.array_equal(a["vit"][0], g["vit"][0]) and np.array_equal(a["vit"][1], g["vit"][1]) and np.array_equal(a["vit"][2], g["vit"][2]),
                  cnt=close(a["cnt"][0], g["cnt"][0], 1e-4, 1e-6))
    if Vo[-1, -1, -1] > -math.inf:
        checks["path"] = np.array_equal(a["vit"][2][a["vit"][1][0]:a["vit"][1][1]], om.traceback(x, y, Vo))
    if not all(checks.values()):
        bad += 1
        print("MISMATCH case %d (seed %d): S=%d nIn=%d nOut=%d pairs=%s kernel=%s -> %s" % (c, seed0 + c, S, nIn, nOut, [(len(p[0]), len(p[1])) for p in pairs], a["kern"], {k: v for k, v in checks.items() if not v}), flush=True)
    b.close() if hasattr(b, "close") else None
    dm.close()
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)

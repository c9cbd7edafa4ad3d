"""Device-free fuzz of the one-tape family's retimed planner: random generators / recognisers x lanes per workgroup x ring in LDS or
L2 x longer periods, the record streams (mb_debug_wide_retimed) replayed by tests/test_retimed_plan.simulate against the oracle --
max (bit for bit), sum, backward sum, and the traceback-code program walked into the oracle's path.
usage: python scripts/fuzz_retimed_plan.py [cases=100] [seed0=7000]"""
import math, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_machine
from test_retimed_plan import simulate, walk_codes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
tmp = tempfile.mkdtemp(); bad = 0; skipped = 0; t0 = time.time()
for c in range(n):
    rng = np.random.RandomState(seed0 + c)
    S = int(rng.choice([4, 9, 25, 40, 70, 120]))
    gen = bool(c % 3)
    nt = int(rng.randint(1, 5))
    em = random_machine(S, 0 if gen else nt, nt if gen else 0, seed0 + c, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.5)), allow_inf=bool(c % 4 == 0))
    for k in ("MB_WIDE_LANES", "MB_WIDE_GLOBAL_VECTORS", "MB_WIDE_RETIMED_PERIOD"): os.environ.pop(k, None)
    if c % 2: os.environ["MB_WIDE_LANES"] = "256" if c % 4 == 1 else "1024"
    if c % 5 == 2: os.environ["MB_WIDE_GLOBAL_VECTORS"] = "1"
    om = oracle.OracleMachine(em); z = np.zeros(0, np.int32)
    try:
        base = capi.debug_wide_retimed(em, tmp + "/b.bin", capi.MB_VITERBI, False)
        if c % 7 == 3: os.environ["MB_WIDE_RETIMED_PERIOD"] = str(base["period"] + int(rng.randint(1, 4)))
        progs = {(capi.MB_VITERBI, False): capi.debug_wide_retimed(em, tmp + "/v.bin", capi.MB_VITERBI, False),
                 (capi.MB_FORWARD, False): capi.debug_wide_retimed(em, tmp + "/f.bin", capi.MB_FORWARD, False),
                 (capi.MB_FORWARD, True): capi.debug_wide_retimed(em, tmp + "/r.bin", capi.MB_FORWARD, True)}
    except capi.MbError as e:
        skipped += 1; continue
    try: tbp = capi.debug_wide_retimed(em, tmp + "/t.bin", capi.MB_VITERBI, False, tb_codes=True)
    except capi.MbError: tbp = None
    ok = True
    for L in (0, int(rng.randint(1, 12)), int(rng.randint(12, 40))):
        seq = rng.randint(1, nt + 1, size=L).astype(np.int32)
        x, y = (z, seq) if gen else (seq, z)
        V = om.viterbi(x, y).reshape(L + 1, S)
        ok &= np.array_equal(simulate(progs[(capi.MB_VITERBI, False)], seq, False, True), V)
        for bwd in (False, True):
            R = (om.backward(x, y, oracle.SUM_EXACT) if bwd else om.forward(x, y, oracle.SUM_EXACT)).reshape(L + 1, S)
            got = simulate(progs[(capi.MB_FORWARD, bwd)], seq, bwd, False); fin = np.isfinite(R)
            ok &= np.array_equal(np.isneginf(got), np.isneginf(R)) and np.allclose(got[fin], R[fin], rtol=1e-11, atol=1e-11)
        if tbp is not None:
            cells, codes = simulate(tbp, seq, False, True, tb=True)
            ok &= np.array_equal(cells, V)
            if V[-1, -1] > -math.inf: ok &= np.array_equal(walk_codes(tbp, codes, L), om.traceback(x, y, om.viterbi(x, y)))
    if not ok: bad += 1; print("MISMATCH case", c, "S", S, "generator" if gen else "recogniser", nt, {k: v for k, v in os.environ.items() if k.startswith("MB_WIDE")}, flush=True)
print("%d cases (%d without a retimed program), %d mismatches, %.1f s" % (n, skipped, bad, time.time() - t0))

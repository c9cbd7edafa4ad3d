"""Device-free fuzz of the tiled family's planner: random machines x random planner knobs x columns per wavefront, the programs
(mb_debug_jit_source mode + 32) replayed by tests/test_tiled_plan.replay against the oracle.  usage: python scripts/fuzz_tiled_plan.py [cases=200] [seed0=5000]"""
import math, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_machine, random_seq
from test_tiled_plan import replay, walk_chosen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
tmp = tempfile.mkdtemp(); bad = 0; t0 = time.time()
for c in range(n):
    rng = np.random.RandomState(seed0 + c)
    S = int(rng.choice([17, 20, 33, 48, 64, 100, 150, 257]))
    nIn, nOut = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    em = random_machine(S, nIn, nOut, seed0 + c, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.5)), allow_inf=bool(c % 4 == 0))
    knobs = {"MB_MEDIUM_SPLIT_DEGREE": str(int(rng.choice([3, 5, 8, 12, 100000]))), "MB_MEDIUM_SPLIT_PART": str(int(rng.choice([2, 4, 8]))),
             "MB_MEDIUM_SYNC_COST": str(int(rng.choice([1, 6]))), "MB_MEDIUM_ROUND_COST": str(int(rng.choice([0, 1, 3]))), "MB_MEDIUM_COUNT_FLAT": str(int(rng.choice([1, 1, 0])))}
    os.environ.update(knobs)
    G = int(rng.choice([1, 2, 4, 8, 16, 32])); K = int(rng.choice([1, 2, 3, 4, 7, 11])); bwd = bool(c % 5 == 1)
    om = oracle.OracleMachine(em)
    pairs = [(random_seq(rng, int(rng.randint(0, 7)), nIn), random_seq(rng, int(rng.randint(0, 9)), nOut)) for _ in range(2)]
    try:
        progF = capi.debug_medium_program(em, tmp + "/f.bin", mode=capi.MB_FORWARD, backward=bwd, closure=K, G=G)
        progV = capi.debug_medium_program(em, tmp + "/v.bin", mode=capi.MB_VITERBI, closure=0, G=G)
    except capi.MbError as e:
        print("case", c, "skipped:", e); continue
    try: progC = capi.debug_medium_program(em, tmp + "/c.bin", mode=3, closure=(K if knobs["MB_MEDIUM_COUNT_FLAT"] == "1" else 0), G=G)
    except capi.MbError: progC = None
    ref_c = np.zeros(em.nTransitions); got_c = np.zeros(em.nTransitions); ok = True
    for x, y in pairs:
        V = om.viterbi(x, y); gotV, chosen = replay(progV, x, y, True, edges=True)
        ok &= np.array_equal(gotV, V)
        if V.reshape(-1)[-1] > -math.inf: ok &= np.array_equal(walk_chosen(em, chosen, x, y), om.traceback(x, y, V))      # the path the program's candidate order chooses
        R = om.backward(x, y, oracle.SUM_EXACT) if bwd else om.forward(x, y, oracle.SUM_EXACT)
        got = replay(progF, x[::-1], y[::-1], False)[::-1, ::-1] if bwd else replay(progF, x, y, False)
        fin = np.isfinite(R)
        ok &= np.array_equal(np.isneginf(got), np.isneginf(R)) and np.allclose(got[fin], R[fin], rtol=1e-10, atol=1e-10)
        if progC is not None and om.loglike(x, y, oracle.SUM_EXACT) > -math.inf:
            ll = om.counts_add(x, y, ref_c, oracle.SUM_EXACT)
            got_c += replay(progC, x, y, False, bwd=om.backward(x, y, oracle.SUM_EXACT), ll=ll, n_trans=em.nTransitions)[1]
    ok &= np.allclose(got_c, ref_c, rtol=1e-8, atol=1e-11)
    if not ok: bad += 1; print("MISMATCH case", c, "S", S, nIn, nOut, "G", G, "K", K, "backward", bwd, knobs, flush=True)
print("%d cases, %d mismatches, %.1f s" % (n, bad, time.time() - t0))

"""Device-free fuzz of the small-machine family's program (mb_debug_small_source mode + 32), replayed by tests/test_small_plan against
the oracle: Viterbi cells bit for bit, traceback bytes walked into the oracle's path, Forward / Backward cells, counts.
usage: python scripts/fuzz_small_plan.py [cases=300] [seed0=3000]"""
import math, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_machine, random_seq
from test_small_plan import replay, walk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tmp = tempfile.mkdtemp(); bad = 0; skipped = 0; t0 = time.time()
for c in range(n):
    rng = np.random.RandomState(seed0 + c)
    S = int(rng.randint(1, 17)); nIn, nOut = int(rng.randint(1, 5)), int(rng.randint(1, 5))
    em = random_machine(S, nIn, nOut, seed0 + c, density=float(rng.uniform(0.8, 3.5)), silent_density=float(rng.uniform(0.0, 2.5)), allow_inf=bool(c % 4 == 0))
    om = oracle.OracleMachine(em)
    try:
        prog = capi.debug_small_program(em, tmp + "/f.bin"); progB = capi.debug_small_program(em, tmp + "/b.bin", backward=True)
    except capi.MbError:
        skipped += 1; continue
    ok = True; ref_c = np.zeros(em.nTransitions); got_c = np.zeros(em.nTransitions)
    for _ in range(2):
        x, y = random_seq(rng, int(rng.randint(0, 8)), nIn), random_seq(rng, int(rng.randint(0, 8)), nOut)
        V = om.viterbi(x, y); cells, first = replay(prog, x, y, "tb")
        ok &= np.array_equal(cells, V)
        if V.reshape(-1)[-1] > -math.inf: ok &= np.array_equal(walk(prog, first, x, y), om.traceback(x, y, V))
        F = om.forward(x, y, oracle.SUM_EXACT); B = om.backward(x, y, oracle.SUM_EXACT)
        for got, ref in ((replay(prog, x, y, "sum"), F), (replay(progB, x[::-1], y[::-1], "sum")[::-1, ::-1], B)):
            fin = np.isfinite(ref)
            ok &= np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)
        if F.reshape(-1)[-1] > -math.inf:
            ll = om.counts_add(x, y, ref_c, oracle.SUM_EXACT)
            got_c += replay(prog, x, y, "count", bwd=B, ll=ll, n_trans=em.nTransitions)[1]
    ok &= np.allclose(got_c, ref_c, rtol=1e-10, atol=1e-13)
    if not ok: bad += 1; print("MISMATCH case", c, "S", S, nIn, nOut, flush=True)
print("%d cases (%d the family does not take), %d mismatches, %.1f s" % (n, skipped, bad, time.time() - t0))

"""Device-free fuzz of the one-tape family's k-workgroups-per-sequence planner: random BLOCK machines (cycles inside blocks
only, so the transition graph has cuts) x k x lanes per part, the parts' record streams (mb_debug_wide_parts) replayed in workgroup order by
tests/test_retimed_plan.simulate_parts against the oracle -- max (bit for bit), sum, backward sum, and the traceback codes walked into the
oracle's paths.   usage: python scripts/fuzz_parts_plan.py [cases=100] [seed0=9000]"""
import math, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_block_machine
from test_retimed_plan import simulate, simulate_parts, walk_codes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
tmp = tempfile.mkdtemp(); bad = 0; skipped = 0; parts_seen = {}; t0 = time.time()
for c in range(n):
    rng = np.random.RandomState(seed0 + c)
    blocks, per = int(rng.choice([2, 3, 5, 8, 12])), int(rng.choice([2, 4, 9, 16]))
    gen = bool(c % 3)
    nt = int(rng.randint(1, 5))
    em = random_block_machine(blocks, per, 0 if gen else nt, nt if gen else 0, seed0 + c, density=float(rng.uniform(0.8, 3.0)),
                              silent_density=float(rng.uniform(0.2, 2.5)), allow_inf=bool(c % 4 == 0))
    S = em.nStates
    k = int(rng.choice([2, 3, 4, 7])); lanes = int(rng.choice([64, 128, 256, 1024]))
    om = oracle.OracleMachine(em); z = np.zeros(0, np.int32)
    try:
        progs = {(capi.MB_VITERBI, False): capi.debug_wide_parts(em, tmp + "/v.bin", k, lanes, capi.MB_VITERBI, False),
                 (capi.MB_FORWARD, False): capi.debug_wide_parts(em, tmp + "/f.bin", k, lanes, capi.MB_FORWARD, False),
                 (capi.MB_FORWARD, True): capi.debug_wide_parts(em, tmp + "/r.bin", k, lanes, capi.MB_FORWARD, True)}
    except capi.MbError as e:
        skipped += 1; continue
    np_ = len(progs[(capi.MB_VITERBI, False)]["parts"]); parts_seen[np_] = parts_seen.get(np_, 0) + 1
    try: tbp = capi.debug_wide_parts(em, tmp + "/t.bin", k, lanes, capi.MB_VITERBI, False, tb_codes=True)
    except capi.MbError: tbp = None
    ok = True
    try:
        for L in (0, int(rng.randint(1, 12)), int(rng.randint(12, 40))):
            seq = rng.randint(1, nt + 1, size=L).astype(np.int32)
            x, y = (z, seq) if gen else (seq, z)
            V = om.viterbi(x, y).reshape(L + 1, S)
            ok &= np.array_equal(simulate_parts(progs[(capi.MB_VITERBI, False)], seq, False, True), V)
            for bwd in (False, True):
                R = (om.backward(x, y, oracle.SUM_EXACT) if bwd else om.forward(x, y, oracle.SUM_EXACT)).reshape(L + 1, S)
                got = simulate_parts(progs[(capi.MB_FORWARD, bwd)], seq, bwd, False); fin = np.isfinite(R)
                ok &= np.array_equal(np.isneginf(got), np.isneginf(R)) and np.allclose(got[fin], R[fin], rtol=1e-11, atol=1e-11)
            if tbp is not None:
                cells, codes = simulate_parts(tbp, seq, False, True, tb=True)
                ok &= np.array_equal(cells, V)
                tabs = {"S": S, "tbOff": tbp["tbOff"], "tbEntry": tbp["tbEntry"], "inEid": tbp["inEid"]}
                if V[-1, -1] > -math.inf: ok &= np.array_equal(walk_codes(tabs, codes, L), om.traceback(x, y, om.viterbi(x, y)))
    except AssertionError as e:
        ok = False; print("ASSERTION", e)
    if not ok: bad += 1; print("MISMATCH case", c, "blocks", blocks, "x", per, "generator" if gen else "recogniser", nt, "k", k, "lanes", lanes, flush=True)
print("%d cases (%d without a cut), parts built %s, %d mismatches, %.1f s" % (n, skipped, dict(sorted(parts_seen.items())), bad, time.time() - t0))

"""GPU fuzz of the one-tape family's k workgroups per sequence: random BLOCK machines (tests/randmachine.random_block_machine),
forced into the one-tape family at any size, random k / lanes per part / sharing of CUs, ragged batches -- against the oracle: Viterbi
matrix of one sequence and every score and path bit for bit (paths through traceback codes and through the fp64 matrix), Forward / Backward
matrices, rolling log-likelihoods (cut in two) and counts within the fast-path tolerance; and against the one-workgroup sweep.
usage: python scripts/fuzz_parts_gpu.py [cases=100] [seed0=11000] [big]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["MB_WIDE_MIN_STATES"] = "1"
os.environ["MB_ONETAPE_TRACEBACK_MIN_TRANS"] = "0"
os.environ["MB_ONETAPE_PARTS_MIN_LEN"] = "0"
os.environ.setdefault("MB_ONETAPE_PART_TIMEOUT_S", "10")
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_block_machine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 11000
REL, ABS = 2e-6, 2e-5
def close(a, b, rel=REL, abs_=ABS):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.shape != b.shape or not np.array_equal(np.isneginf(a), np.isneginf(b)): return False
    f = np.isfinite(b)
    return bool(np.all(np.abs(a[f] - b[f]) <= abs_ + rel * np.abs(b[f])))
bad = 0; ran = 0; with_parts = 0; t0 = time.time()
for c in range(n):
    rng = np.random.RandomState(seed0 + c)
    big = len(sys.argv) > 3 and sys.argv[3] == "big"      # machines of up to 2 400 states (relays, several slots per round)
    blocks, per = (int(rng.choice([12, 20, 40])), int(rng.choice([25, 40, 60]))) if big else (int(rng.choice([2, 3, 5, 8, 12, 20])), int(rng.choice([3, 6, 12, 25])))
    gen = bool(c % 3)
    nt = int(rng.randint(1, 5))
    em = random_block_machine(blocks, per, 0 if gen else nt, nt if gen else 0, seed0 + c, density=float(rng.uniform(0.8, 3.0)),
                              silent_density=float(rng.uniform(0.2, 2.5)), allow_inf=bool(c % 4 == 0))
    k = int(rng.choice([2, 3, 4, 7, 16])); lanes = int(rng.choice([64, 128, 256, 512]))
    os.environ["MB_ONETAPE_PARTS"] = str(k); os.environ["MB_ONETAPE_PART_LANES"] = str(lanes)
    os.environ["MB_ONETAPE_PART_EXCLUSIVE"] = "0" if c % 5 == 0 else "1"
    om = oracle.OracleMachine(em); z = np.zeros(0, np.int32)
    lens = [0, int(rng.randint(1, 10)), int(rng.randint(10, 90)), int(rng.randint(60, 200))][: int(rng.randint(2, 5))]
    seqs = [rng.randint(1, nt + 1, size=L).astype(np.int32) for L in lens]
    pairs = [(z, q) if gen else (q, z) for q in seqs]
    ok = True; names = []
    try:
        dm = capi.DeviceMachine(em)
        x, y = pairs[-1]
        V = dm.fill(capi.MB_VITERBI, x, y); names.append(capi.last_kernel_name())
        F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
        ok &= np.array_equal(V, om.viterbi(x, y)) and close(F, om.forward(x, y, oracle.SUM_EXACT)) and close(B, om.backward(x, y, oracle.SUM_EXACT))
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        ll = b.forward(capi.MB_ROLLING); names.append(capi.last_kernel_name())
        vll, off, edges = b.viterbi(); names.append(capi.last_kernel_name())
        os.environ["MB_ONETAPE_TB"] = "0"
        vll2, off2, edges2 = b.viterbi()
        os.environ.pop("MB_ONETAPE_TB")
        counts, s, _ = b.counts()
        ref = np.zeros(em.nTransitions)
        for q, (xx, yy) in enumerate(pairs):
            Vo = om.viterbi(xx, yy)
            ok &= vll[q] == Vo[-1, -1, -1] and vll2[q] == vll[q] and close([ll[q]], [om.loglike(xx, yy, oracle.SUM_EXACT)])
            if Vo[-1, -1, -1] > -math.inf:
                tb = om.traceback(xx, yy, Vo)
                ok &= np.array_equal(edges[off[q]:off[q + 1]], tb) and np.array_equal(edges2[off2[q]:off2[q + 1]], tb)
            if om.forward(xx, yy, oracle.SUM_EXACT)[-1, -1, -1] > -math.inf: om.counts_add(xx, yy, ref, oracle.SUM_EXACT)
        ok &= close(counts, ref, 1e-5, 1e-7)
        dm.close()
        ran += 1; with_parts += any(" parts" in nm for nm in names)
    except Exception as e:
        ok = False; print("EXCEPTION", type(e).__name__, str(e)[:200])
    if not ok: bad += 1; print("MISMATCH case", c, "blocks", blocks, "x", per, "generator" if gen else "recogniser", nt, "k", k, "lanes", lanes, names, flush=True)
print("%d cases, %d through k workgroups per sequence, %d mismatches, %.1f s" % (ran, with_parts, bad, time.time() - t0))

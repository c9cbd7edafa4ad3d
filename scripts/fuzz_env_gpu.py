"""Randomised parity sweep of RESTRICTED ENVELOPES on the GPU: random two-tape machines (all kernel families by state count),
ragged batches in which some pairs carry the path-area envelope of a random alignment (random width, 0 = the path
itself) and the others the full one; automatically chosen family against the generic family (fills, rolling
log-likelihood, Viterbi paths, counts) and against the C oracle for the first pair.
usage: python scripts/fuzz_env_gpu.py [cases] [seed0]"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.seqpair import Envelope
from oracle import oracle
from randmachine import random_machine, random_seq

def close(a, b, rel, abs_):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ok = (np.isneginf(a) & np.isneginf(b)) | (np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= abs_ + rel * np.abs(b)))
    return bool(np.all(ok))

def random_envelope(rng, il, ol):
    cols = []; i = o = 0
    while i < il or o < ol:
        moves = [m for m in ("d", "i", "o") if (m != "d" or (i < il and o < ol)) and (m != "i" or i < il) and (m != "o" or o < ol)]
        mv = moves[int(rng.randint(len(moves)))]
        if mv == "d": cols.append(("a", "b")); i += 1; o += 1
        elif mv == "i": cols.append(("a", "")); i += 1
        else: cols.append(("", "b")); o += 1
    return Envelope.pathAreaEnvelope(cols, int(rng.choice([0, 1, 2, 3, 5, 9, 20])))

def make_case(seed):
    """(machine, pairs, envelopes) of one case: everything is drawn from RandomState(seed)."""
    c = seed % 1000
    rng = np.random.RandomState(seed)
    S = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 17, 33, 64, 100, 257]))
    nIn = int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
    em = random_machine(S, nIn, nOut, seed, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=bool(c % 5 == 0))
    n = int(rng.randint(1, 6))
    scale = 5 if (c % 4 == 1 and S <= 64) or (c % 4 == 3 and S <= 100) else 1   # longer lattices: several strips and tiles, most of them outside the band
    pairs = [(random_seq(rng, int(rng.randint(0, 40 * scale)), nIn), random_seq(rng, int(rng.randint(0, 60 * scale)), nOut)) for _ in range(n)]
    envs = [random_envelope(rng, len(x), len(y)) if (k == 0 or rng.rand() < 0.6) else None for k, (x, y) in enumerate(pairs)]
    return em, pairs, envs


def run_case(seed, verbose=True):
    """Runs one case; returns the dict of failed checks (empty = parity)."""
    em, pairs, envs = make_case(seed)
    dm = capi.DeviceMachine(em); om = oracle.OracleMachine(em)
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    b.set_envelopes([(e.inStart, e.inEnd) if e is not None else None for e in envs])
    out = {}
    x, y = pairs[0]; e0 = envs[0]
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
        capi.set_kernel(fam)
        try:
            out[fam] = dict(F=dm.fill(capi.MB_FORWARD, x, y, 0, e0.inStart, e0.inEnd), B=dm.fill(capi.MB_BACKWARD, x, y, 0, e0.inStart, e0.inEnd),
                            V=dm.fill(capi.MB_VITERBI, x, y, 0, e0.inStart, e0.inEnd), kern=capi.last_kernel_name(),
                            ll=b.forward(capi.MB_ROLLING), llm=b.forward(capi.MB_MATERIALISE), vit=b.viterbi(), cnt=b.counts())
        finally:
            capi.set_kernel(capi.KERNEL_AUTO)
    a, g = out[capi.KERNEL_AUTO], out[capi.KERNEL_GENERIC]
    with oracle.envelope(e0.inStart, e0.inEnd):
        Vo = om.viterbi(x, y); Fo = om.forward(x, y, oracle.SUM_EXACT)
        po = om.traceback(x, y, Vo) if Vo[-1, -1, -1] > -math.inf else None
    checks = dict(V=np.array_equal(a["V"], g["V"]) and np.array_equal(a["V"], Vo),
                  F=close(a["F"], g["F"], 2e-6, 2e-5) and close(a["F"], Fo, 2e-6, 2e-5),
                  B=close(a["B"], g["B"], 2e-6, 2e-5), ll=close(a["ll"], g["ll"], 2e-6, 2e-5) and close(a["ll"], a["llm"], 1e-9, 1e-12),
                  vit=np.array_equal(a["vit"][0], g["vit"][0]) and np.array_equal(a["vit"][1], g["vit"][1]) and np.array_equal(a["vit"][2], g["vit"][2]),
                  cnt=close(a["cnt"][0], g["cnt"][0], 1e-4, 1e-6) and close(a["cnt"][1], g["cnt"][1], 2e-6, 2e-5) and close(a["cnt"][2], g["cnt"][2], 2e-6, 2e-5))
    if po is not None:
        checks["path"] = np.array_equal(a["vit"][2][a["vit"][1][0]:a["vit"][1][1]], po)
    failed = {k: v for k, v in checks.items() if not v}
    if failed and verbose:
        print("MISMATCH seed %d: S=%d nIn=%d nOut=%d pairs=%s env=%s kernel=%s -> %s" % (seed, em.nStates, em.nInTok, em.nOutTok, [(len(p[0]), len(p[1])) for p in pairs], [e is not None for e in envs], a["kern"], failed), flush=True)
    dm.close()
    return failed


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t0 = time.time()
    for c in range(cases):
        bad += 1 if run_case(seed0 + c) else 0
    print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)

"""Tiled family, the modes that keep no fp64 Forward / Viterbi matrix (round 3): Viterbi with ONE traceback byte per cell
(MED_MODE_TB + k_traceback_bytes) and the count sweep without a Forward matrix (MED_MAT_ROLL), against the oracle and
against the round-2 paths (MB_MEDIUM_TB=0, MB_MEDIUM_COUNTS_ROLL=0) -- parity first, then timings on psw2dna 64 x 487 x 2 kb.
usage: python scripts/check_tiled_modes.py [parity|time|all]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch, synth_tokens
from oracle import oracle
from randmachine import random_machine, random_seq

what = sys.argv[1] if len(sys.argv) > 1 else "all"


def close(a, b, rel, abs_):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return bool(np.all((np.isneginf(a) & np.isneginf(b)) | (np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= abs_ + rel * np.abs(b)))))


def parity_case(em, pairs, tag):
    om = oracle.OracleMachine(em)
    res = {}
    for tbk, roll in (("1", "1"), ("0", "0")):
        capi.set_option("MB_MEDIUM_TB", tbk); capi.set_option("MB_MEDIUM_COUNTS_ROLL", roll)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        v = b.viterbi(); kv = capi.last_kernel_name()
        c = b.counts(); kc = capi.last_kernel_name()
        res[tbk] = (v, c, kv, kc)
        dm.close()
    capi.set_option("MB_MEDIUM_TB", None); capi.set_option("MB_MEDIUM_COUNTS_ROLL", None)
    (v1, c1, kv, kc), (v0, c0, _, _) = res["1"], res["0"]
    ok = np.array_equal(v1[0], v0[0]) and np.array_equal(v1[1], v0[1]) and np.array_equal(v1[2], v0[2])
    okc = close(c1[0], c0[0], 1e-6, 1e-9) and close(c1[2], c0[2], 1e-9, 1e-9)
    # oracle: first two pairs
    oko = True
    ref_c = np.zeros(em.nTransitions)
    for k, (x, y) in enumerate(pairs[:2]):
        V = om.viterbi(x, y)
        oko = oko and v1[0][k] == V[-1, -1, -1]
        if V[-1, -1, -1] > -math.inf:
            oko = oko and np.array_equal(v1[2][v1[1][k]:v1[1][k + 1]], om.traceback(x, y, V))
    print("%-34s viterbi new==old %s  ==oracle %s   counts new~old %s   [%s | %s]" % (tag, ok, oko, okc, kv, kc), flush=True)
    return ok and oko and okc


if what in ("parity", "all"):
    good = True
    m = Machine.fromFile(os.path.join(ROOT, "tests/golden/preset/psw2dna.json")); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    good &= parity_case(em, [synth_tokens(k + 1, il, ol, em.nInTok, em.nOutTok) for k, (il, ol) in enumerate([(30, 90), (100, 300), (0, 5), (7, 0), (33, 257), (64, 129)])], "psw2dna")
    for S, seed in [(17, 1), (24, 2), (40, 3), (64, 4), (100, 5), (150, 6), (257, 7), (300, 8), (33, 9), (48, 10)]:
        rng = np.random.RandomState(seed)
        em = random_machine(S, 3, 2, 100 + seed, density=2.0, silent_density=1.0, dup=True)
        pairs = [(random_seq(rng, int(rng.randint(0, 90)), 3), random_seq(rng, int(rng.randint(0, 200)), 2)) for _ in range(5)]
        good &= parity_case(em, pairs, "random S=%d" % S)
    print("PARITY", "OK" if good else "FAILED")

if what in ("time", "all"):
    m = Machine.fromFile(os.path.join(ROOT, "tests/golden/preset/psw2dna.json")); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    for n, il, ol in ((64, 487, 2000), (32, 487, 10000)):
        for tbk, roll in (("0", "0"), ("1", "1")):
            capi.set_option("MB_MEDIUM_TB", tbk); capi.set_option("MB_MEDIUM_COUNTS_ROLL", roll)
            dm = capi.DeviceMachine(em)
            b = capi.DeviceBatch(dm, *synth_batch(4, n, il, ol, em.nInTok, em.nOutTok))
            cells = b.cells()
            for name, fn in (("viterbi+paths", lambda: b.viterbi(paths=True)), ("viterbi fill", lambda: b.viterbi(paths=False)), ("counts", lambda: b.counts())):
                fn(); ms = []
                for _ in range(3):
                    t0 = time.perf_counter(); fn(); wall = (time.perf_counter() - t0) * 1e3
                    ms.append((capi.last_device_ms(), wall))
                dev = min(x[0] for x in ms); wall = min(x[1] for x in ms)
                print("%d x %d x %d  TB/ROLL=%s  %-14s device %8.2f ms  wall %8.2f ms  %7.1f G cells/s (device)  kernel %s launches %d" %
                      (n, il, ol, tbk, name, dev, wall, cells / dev / 1e6, capi.last_kernel_name(), capi.last_launch_count()), flush=True)
            dm.close()
    capi.set_option("MB_MEDIUM_TB", None); capi.set_option("MB_MEDIUM_COUNTS_ROLL", None)

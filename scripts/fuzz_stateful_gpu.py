"""Stateful parity sweep on the GPU: ONE DeviceMachine and ONE DeviceBatch per case live through a random sequence of
operations -- weight updates, envelope changes (restricted / full again), memory-budget changes that force chunking, and the
DP calls in between -- and every result is compared with a FRESH machine + batch run through the generic family with the
same weights and envelopes.  Exercises what the one-shot fuzzers do not: cached tile lists, refreshed weight tables of the
run-time specialised kernels, rebuilt one-tape programs, recycled workspaces.
usage: python scripts/fuzz_stateful_gpu.py [cases] [seed0]"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
from machineboss_amd import capi
from randmachine import random_machine, random_seq
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_env_gpu", os.path.join(ROOT, "scripts", "fuzz_env_gpu.py"))
fe = importlib.util.module_from_spec(spec); spec.loader.exec_module(fe)
close = fe.close


def reference(em, lw, pairs, envs, what):
    """the same call on fresh objects through the generic family"""
    capi.set_kernel(capi.KERNEL_GENERIC)
    try:
        dm = capi.DeviceMachine(em); dm.set_weights(lw)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        if any(e is not None for e in envs):
            b.set_envelopes([(e.inStart, e.inEnd) if e is not None else None for e in envs])
        r = {"rolling": lambda: b.forward(capi.MB_ROLLING), "mat": lambda: b.forward(capi.MB_MATERIALISE), "viterbi": lambda: b.viterbi(), "counts": lambda: b.counts()}[what]()
        dm.close()
        return r
    finally:
        capi.set_kernel(capi.KERNEL_AUTO)


def same(what, a, g):
    if what in ("rolling", "mat"):
        return close(a, g, 2e-6, 2e-5)
    if what == "viterbi":
        return np.array_equal(a[0], g[0]) and np.array_equal(a[1], g[1]) and np.array_equal(a[2], g[2])
    return close(a[0], g[0], 1e-4, 1e-6) and close(a[1], g[1], 2e-6, 2e-5) and close(a[2], g[2], 2e-6, 2e-5)


def run_case(seed):
    rng = np.random.RandomState(seed)
    oneTape = seed % 4 == 3
    S = int(rng.choice([2, 5, 8, 12, 17, 40, 100, 260])) if not oneTape else int(rng.choice([30, 260, 400]))
    nIn = 0 if oneTape else int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
    em = random_machine(S, nIn, nOut, seed, density=float(rng.uniform(0.8, 2.5)), silent_density=float(rng.uniform(0.2, 1.5)), allow_inf=bool(seed % 5 == 0))
    scale = 4 if S <= 40 else 1
    n = int(rng.randint(2, 7))
    pairs = [(random_seq(rng, int(rng.randint(0, 40 * scale)) if nIn else 0, nIn), random_seq(rng, int(rng.randint(1, 60 * scale)), nOut)) for _ in range(n)]
    lw = np.array(em.logWeight, dtype=np.float64)
    envs = [None] * n
    dm = capi.DeviceMachine(em); b = capi.DeviceBatch.from_pairs(dm, pairs)
    bad = []
    try:
        for step in range(10):
            op = rng.choice(["weights", "env", "budget", "call", "call", "call"]) if step else "call"
            if op == "weights":
                fin = np.isfinite(lw)
                lw = lw.copy(); lw[fin] += rng.uniform(-0.7, 0.3, size=int(fin.sum()))
                dm.set_weights(lw)
            elif op == "env" and not oneTape:
                envs = [fe.random_envelope(rng, len(x), len(y)) if rng.rand() < 0.5 else None for x, y in pairs]
                if all(e is None for e in envs) and rng.rand() < 0.5:
                    envs[0] = fe.random_envelope(rng, len(pairs[0][0]), len(pairs[0][1]))
                b.set_envelopes([(e.inStart, e.inEnd) if e is not None else None for e in envs])   # all None: full envelopes again
            elif op == "budget":
                one = max((len(x) + 1) * (len(y) + 1) * S * 8 for x, y in pairs)
                capi.set_memory_budget(int(rng.choice([0, 4 * one, 9 * one])))
            else:
                what = str(rng.choice(["rolling", "mat", "viterbi", "counts"]))
                call = {"rolling": lambda: b.forward(capi.MB_ROLLING), "mat": lambda: b.forward(capi.MB_MATERIALISE), "viterbi": lambda: b.viterbi(), "counts": lambda: b.counts()}[what]
                try:
                    got = call()
                except capi.MbError as e:      # a budget below one (padded) matrix is refused, not chunked: lift it and go on
                    if "exceeds the device memory budget" not in str(e): raise
                    capi.set_memory_budget(0)
                    got = call()
                kern = capi.last_kernel_name()
                ref = reference(em, lw, pairs, envs, what)
                if not same(what, got, ref):
                    bad.append((step, what, kern))
    finally:
        capi.set_memory_budget(0)
        dm.close()
    if bad:
        print("MISMATCH seed %d: S=%d nIn=%d nOut=%d pairs=%s -> %s" % (seed, S, nIn, nOut, [(len(x), len(y)) for x, y in pairs], bad), flush=True)
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); nbad = 0
    for c in range(cases):
        nbad += 1 if run_case(seed0 + c) else 0
    print("%d cases, %d mismatches, %.1f s" % (cases, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)

"""Small one-tape machines (random generators / recognisers of S states): which family should take them?
usage: python scripts/bench_small_onetape.py S [pairs len]   (MB_WIDE_MIN_STATES from the environment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from randmachine import random_machine, random_seq
S = int(sys.argv[1]); pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256; L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
for tape in ("out", "in"):
    em = random_machine(S, 0 if tape == "out" else 3, 3 if tape == "out" else 0, 5, density=2.0, silent_density=1.2)
    dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(1)
    z = np.zeros(0, np.int32)
    ps = [(z, random_seq(rng, L, 3)) if tape == "out" else (random_seq(rng, L, 3), z) for _ in range(pairs)]
    b = capi.DeviceBatch.from_pairs(dm, ps)
    cells = b.cells()
    for name, f in (("forward rolling", lambda: b.forward(capi.MB_ROLLING)), ("viterbi + paths", lambda: b.viterbi()), ("counts", lambda: b.counts())):
        f(); t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
        print("S=%d tape=%s %-16s %8.3f Gcells/s %8.2f ms  %s" % (S, tape, name, cells / dt / 1e9, dt * 1e3, capi.last_kernel_name()), flush=True)

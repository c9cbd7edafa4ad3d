"""One pair through the batch entry points: wall and device time per call, launch by launch against the one-launch sweep
(MB_SMALL_ONE_LAUNCH).  python scripts/single_pair_probe.py [preset] [inLen] [outLen] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch

preset = sys.argv[1] if len(sys.argv) > 1 else "dnapsw"
il = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ol = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", preset + ".json")), None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(2, 1, il, ol, em.nInTok, em.nOutTok))
ref = {}
for one in ("0", "1", "2"):
    capi.set_option("MB_SMALL_ONE_LAUNCH", one)
    for name, fn in (("forward", lambda: b.forward(capi.MB_ROLLING)), ("viterbi", lambda: b.viterbi(paths=False)[0]), ("align", lambda: b.viterbi(paths=True)[0])):
        r = fn(); r = fn()
        capi.synchronize()
        dev = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn(); dev += capi.last_device_ms()
        capi.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        v = float(np.asarray(r).ravel()[0])
        same = ref.setdefault(name, v) == v
        print("one_launch=%s %-8s %8.3f ms wall %8.3f ms device  %7.1f pairs/s  launches %s  %s  same=%s" % (one, name, wall, dev / reps, 1e3 / wall, capi.last_launch_count(), capi.last_kernel_name(), same))

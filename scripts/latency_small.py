"""Config 1 (one 50-aa protpsw pair, `boss --loglike` plumbing): per-call latency of the batched entry points on a resident batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
m = Machine.fromFile("tests/golden/preset/protpsw.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
for pairs in (1, 16):
    b = capi.DeviceBatch(dm, *synth_batch(1, pairs, 50, 50, em.nInTok, em.nOutTok))
    for name, f in (("forward rolling", lambda: b.forward(capi.MB_ROLLING)), ("forward materialised", lambda: b.forward(capi.MB_MATERIALISE)),
                    ("viterbi + path", lambda: b.viterbi()), ("counts", lambda: b.counts())):
        f(); f()
        t0 = time.perf_counter()
        for _ in range(50): f()
        dt = (time.perf_counter() - t0) / 50
        print("%2d pair(s) %-22s %8.1f us per call (device %.1f us)  %s" % (pairs, name, dt * 1e6, capi.last_device_ms() * 1e3, capi.last_kernel_name()), flush=True)

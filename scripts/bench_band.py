"""Banded fills: dnapsw, pairs x 1 kb, path-area envelope of the given width around the main diagonal, against the full
rectangle (judge item: envelopes on the fast path).  usage: python scripts/bench_band.py [pairs=1024] [len=1000] [width=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
from machineboss_amd.seqpair import Envelope
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
width = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = Machine.fromFile("tests/golden/preset/dnapsw.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
tok = synth_batch(2, pairs, L, L, em.nInTok, em.nOutTok)
full = capi.DeviceBatch(dm, *tok); band = capi.DeviceBatch(dm, *tok)
env = Envelope.pathAreaEnvelope([("a", "b")] * L, width)
band.set_envelopes([(env.inStart, env.inEnd)] * pairs)
frac = float(np.sum(np.asarray(env.inEnd) - np.asarray(env.inStart))) / ((L + 1) * (L + 1))
def t(f):
    f(); held = []; t0 = time.perf_counter()
    for _ in range(3): held.append(f())
    return (time.perf_counter() - t0) / 3, capi.last_device_ms(), capi.last_kernel_name()
for name, fn in (("forward rolling", lambda b: b.forward(capi.MB_ROLLING)), ("forward materialised", lambda b: b.forward(capi.MB_MATERIALISE)),
                 ("viterbi + traceback", lambda b: b.viterbi(paths=True)), ("counts", lambda b: b.counts())):
    tf = t(lambda: fn(full)); tb = t(lambda: fn(band))
    print("%-22s full %7.2f ms (dev %6.2f)  band %7.2f ms (dev %6.2f) = %.3f of full; envelope holds %.3f of the cells -> %.2fx the ideal  %s"
          % (name, tf[0] * 1e3, tf[1], tb[0] * 1e3, tb[1], tb[1] / tf[1], frac, tb[1] / tf[1] / frac, tb[2]), flush=True)

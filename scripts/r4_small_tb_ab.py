"""Device time of the small family's traceback (Viterbi with paths minus the fill) on config 2 and on protpsw 1024 x 400 x 400."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
for preset, n, il, ol in (("dnapsw", 1024, 1000, 1000), ("protpsw", 1024, 400, 400)):
    em = EvaluatedMachine.fromMachine(Machine.fromFile("tests/golden/preset/%s.json" % preset), None, useDefaults=True)
    dm = capi.DeviceMachine(em); b = capi.DeviceBatch(dm, *synth_batch(2, n, il, ol, em.nInTok, em.nOutTok))
    wp, wf = [], []
    for _ in range(6):
        r = b.viterbi(paths=True); wp.append(capi.last_device_ms())
        b.viterbi(paths=False); wf.append(capi.last_device_ms())
    print("%s %d x %d x %d: with paths %.3f ms, fill %.3f ms, traceback %.3f ms (device, best of 6), %d path edges" % (preset, n, il, ol, min(wp), min(wf), min(wp) - min(wf), len(r[2])), flush=True)

"""Host-side timing of repeated mb_batch_viterbi calls on BASELINE config 2 (looks for sporadic stalls).
usage: python scripts/vit_timing.py [keep|drop] [paths|nopaths] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
keep = (sys.argv[1] if len(sys.argv) > 1 else "drop") == "keep"
paths = (sys.argv[2] if len(sys.argv) > 2 else "paths") == "paths"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = Machine.fromFile("tests/golden/preset/dnapsw.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(2, 1024, 1000, 1000, em.nInTok, em.nOutTok))
held = []
for r in range(reps):
    t0 = time.perf_counter(); v = b.viterbi(paths=paths); dt = time.perf_counter() - t0
    if keep:
        held.append(v)
    print("rep", r, "%.2f ms" % (dt * 1e3), "dev %.2f" % capi.last_device_ms(), file=sys.stderr, flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
preset, pairs, il, ol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = Machine.fromFile("tests/golden/preset/%s.json" % preset); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(2, pairs, il, ol, em.nInTok, em.nOutTok))
cells = b.cells()
def t(f, name):
    f(); t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print("%-22s %8.2f Gcells/s  dev %.1f ms  %s" % (name, cells / dt / 1e9, capi.last_device_ms(), capi.last_kernel_name()), flush=True); return r
ll = t(lambda: b.forward(capi.MB_ROLLING), "forward rolling")
llm = t(lambda: b.forward(capi.MB_MATERIALISE), "forward materialised")
v = t(lambda: b.viterbi(paths=False), "viterbi fill")
vp = t(lambda: b.viterbi(paths=True), "viterbi + traceback")
c = t(lambda: b.counts(), "fwd+bwd+counts")
print("checks", float(ll.sum()), float(llm.sum()), float(v[0].sum()), float(c[1]))

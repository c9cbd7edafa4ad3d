"""What a weight update costs the one-tape programs inside the E-step that follows (config 5's machine, 64 sequences): the EM loop's
device-facing part with one workgroup per sequence and with k parts (MB_ONETAPE_PARTS_MIN_LEN=0).  usage: train_replan_probe.py [length]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.seqgen import synth_batch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
m5 = A.composeLeftToRight([HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(20).machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
params = m5.getParamDefs(True)
ev = EvaluatedMachine.fromMachine(m5, params)
dm = capi.DeviceMachine(ev)
b = capi.DeviceBatch(dm, *synth_batch(5, 64, 0, L, ev.nInTok, ev.nOutTok))
b.counts(); t0 = time.perf_counter(); b.counts(); plain = (time.perf_counter() - t0) * 1e3
rows = []
for it in range(4):
    params = {k: (v * (1.0 - 0.01 * (it + 1)) + 0.005 * (it + 1) if isinstance(v, float) and 0.0 < v < 1.0 else v) for k, v in params.items()}
    ev = ev.reweighted(m5, params)
    t0 = time.perf_counter(); dm.set_weights(ev.logWeight); c, s, _ = b.counts(); rows.append((time.perf_counter() - t0) * 1e3)
print("L=%d  plain E-step %.1f ms; set_weights + E-step after a weight update: %s ms  (%s)" % (L, plain, " ".join("%.1f" % r for r in rows), capi.last_kernel_name()))

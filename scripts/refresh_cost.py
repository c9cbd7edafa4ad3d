"""Cost of a weight update on a one-tape machine (what every EM iteration pays before its E-step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
if len(sys.argv) > 1 and not sys.argv[1].isdigit():      # a preset by name (two tapes): 64 pairs x 487 x 2000 (psw2dna), 1024 x 400 x 400 otherwise
    em = EvaluatedMachine.fromMachine(P(sys.argv[1]), None, useDefaults=True)
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch(dm, *(synth_batch(4, 64, 487, 2000, em.nInTok, em.nOutTok) if sys.argv[1] == "psw2dna" else synth_batch(3, 1024, 400, 400, em.nInTok, em.nOutTok)))
else:
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
    em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch(dm, *synth_batch(5, 64, 0, 2000, em.nInTok, em.nOutTok))
t0 = time.perf_counter(); b.counts(); print("first E-step (programs built): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); b.counts(); print("E-step: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
lw = np.array(em.logWeight, dtype=np.float64)
for it in range(3):
    lw = lw - 0.01
    t0 = time.perf_counter(); dm.set_weights(lw); t1 = time.perf_counter(); b.counts(); t2 = time.perf_counter()
    print("weight update %.1f ms, E-step after it %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)

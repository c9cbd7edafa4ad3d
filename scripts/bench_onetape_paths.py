"""Config 5 machine: Viterbi with the paths on the host (fill + traceback + copy), against the fill alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
nodes, pairs, outlen = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 64, 2000)
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(5, pairs, 0, outlen, em.nInTok, em.nOutTok))
cells = b.cells()
for name, f in (("fill", lambda: b.viterbi(paths=False)), ("fill + paths", lambda: b.viterbi())):
    f(); t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print("%-14s %8.2f Gcells/s  %.1f ms (device %.1f ms)  %s  %s" % (name, cells / dt / 1e9, dt * 1e3, capi.last_device_ms(), capi.last_kernel_name(), "" if r[2] is None else "%d path transitions" % len(r[2])), flush=True)

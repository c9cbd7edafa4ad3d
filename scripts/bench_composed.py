"""Config 4b: protpsw . translate . dnapsw (dnapsw's constraints cleared) assembled on the box, Forward
materialised + rolling + Viterbi on `pairs` x 487 aa x `outlen` nt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.machine import Machine, Constraints
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
outlen = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
d = P("dnapsw"); d.cons = Constraints()
t = time.perf_counter(); m = A.compose(A.compose(P("protpsw"), P("translate")), d)
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
print("composed %d states %d transitions in %.2f s" % (em.nStates, em.nTransitions, time.perf_counter() - t))
dm = capi.DeviceMachine(em)
inTok, inOff, outTok, outOff = synth_batch(4, pairs, 487, outlen, em.nInTok, 3)   # DNA over {A,C,G}: no stop codons
b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
cells = b.cells()
def tm(f, name):
    f(); t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print("%-22s %8.2f Gcells/s  dev %.1f ms  %s" % (name, cells / dt / 1e9, capi.last_device_ms(), capi.last_kernel_name()), flush=True); return r
llr = tm(lambda: b.forward(capi.MB_ROLLING), "forward rolling")
llm = tm(lambda: b.forward(capi.MB_MATERIALISE), "forward materialised")
v = tm(lambda: b.viterbi(paths=False), "viterbi fill")
print("finite", np.isfinite(llm).all(), float(llm[0]), float(llr[0]), float(v[0][0]))

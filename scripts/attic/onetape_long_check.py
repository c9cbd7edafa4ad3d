"""One-tape family against the generic family (one __syncthreads() per silent level, exact fp64 log1p/exp) at a length the
CPU oracle would need minutes for: 20-node config-5 machine, 4 x 3000 nt.  Viterbi scores must be identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(5, 4, 0, 3000, em.nInTok, em.nOutTok))
res = {}
for fam, name in ((capi.KERNEL_AUTO, "one-tape"), (capi.KERNEL_GENERIC, "generic")):
    capi.set_kernel(fam)
    t0 = time.perf_counter()
    res[name] = (b.forward(capi.MB_ROLLING), b.forward(capi.MB_MATERIALISE), b.viterbi(paths=True))
    print(name, "%.2f s" % (time.perf_counter() - t0), capi.last_kernel_name(), flush=True)
capi.set_kernel(capi.KERNEL_AUTO)
a, g = res["one-tape"], res["generic"]
print("forward rel diff", float(np.max(np.abs(a[0] - g[0]) / np.abs(g[0]))), float(np.max(np.abs(a[1] - g[1]) / np.abs(g[1]))))
print("viterbi equal", bool(np.array_equal(a[2][0], g[2][0])), "paths equal", bool(np.array_equal(a[2][2], g[2][2])), "path edges", len(a[2][2]))
assert np.array_equal(a[2][0], g[2][0]) and np.array_equal(a[2][2], g[2][2]) and np.max(np.abs(a[0] - g[0]) / np.abs(g[0])) < 1e-6
print("OK")

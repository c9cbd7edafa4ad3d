"""E-step (mb_batch_counts) of BASELINE config 5's machine: wall and device ms, lattice cells/s, the pool's state.
usage: estep_probe.py [nSeq] [length] [reps]      (environment: MB_MEM_FRACTION, MB_TIMING, ... as for the library)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from machineboss_amd import capi, algebra as A
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.seqgen import synth_batch

nSeq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
h = HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(20)
em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
capi.set_device(0)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(5, nSeq, 0, L, em.nInTok, em.nOutTok))
lat = nSeq * (L + 1) * em.nStates
t0 = time.perf_counter(); c, s, _ = b.counts(); print("first call %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for _ in range(reps):
    t0 = time.perf_counter(); c, s, _ = b.counts(); dt = time.perf_counter() - t0
    print("counts %9.2f ms wall %9.2f ms device %6.2f G lattice-cells/s  loglike %.10f  counts sum %.6f  %s" % (dt * 1e3, capi.last_device_ms(), lat / dt / 1e9, s, float(np.sum(c)), capi.last_kernel_name()), flush=True)
print(capi.alloc_stats())

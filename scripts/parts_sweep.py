"""Sweep of the one-tape parts' knobs (k, lanes per part, ring depth) on config 5's machine: ms per call for each mode.
usage: parts_sweep.py nodes pairs outlen modes k1,k2,.. lanes1,lanes2,.. [rings=8,4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
nodes, pairs, outlen, modes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ks = [int(x) for x in sys.argv[5].split(",")]; lanes = [int(x) for x in sys.argv[6].split(",")]
rings = [int(x) for x in (sys.argv[7] if len(sys.argv) > 7 else "8").split(",")]
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(5, pairs, 0, outlen, em.nInTok, em.nOutTok))
def run(mode):
    if mode == "r": return b.forward(capi.MB_ROLLING)
    if mode == "v": return b.viterbi(paths=False)[0]
    if mode == "p": return b.viterbi()[0]
    if mode == "c": return b.counts()[1]
print("%d states, %d sequences x %d; ms per call" % (em.nStates, pairs, outlen))
print("%-22s" % "k, lanes, ring" + "".join("%10s" % m for m in modes))
for k in ks:
    for L in lanes:
        for R in rings:
            os.environ["MB_ONETAPE_PARTS"] = str(k); os.environ["MB_ONETAPE_PART_LANES"] = str(L); os.environ["MB_ONETAPE_PART_RING"] = str(R)
            row = []
            for mode in modes:
                try:
                    run(mode); t0 = time.perf_counter(); run(mode); row.append("%8.1f%s" % ((time.perf_counter() - t0) * 1e3, "*" if " parts" in capi.last_kernel_name() or mode == "c" else " "))
                except Exception as e: row.append("   error ")
            print("%-22s" % ("%d, %d, %d" % (k, L, R)) + "".join("%10s" % x for x in row), flush=True)
            if k == 1: break
        if k == 1: break

"""extra.nonuniform.forward_materialised fell 566 -> 5.92 G cells/s of WALL with the device time unchanged.
Reproduces the bench's order (config-5 block with k workgroups per sequence, then a materialised Forward of psw2dna over 32 pairs
that exceeds the budget) and prints mb_alloc_stats() around every call.  MB_POOL_STICKY=0 restores round 5's budget."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from machineboss_amd import capi, algebra as A
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.seqgen import synth_batch

def call(label, fn):
    a0 = capi.alloc_stats(); t0 = time.perf_counter()
    try: r = fn()
    except Exception as e: r = None; print("   FAILED:", e)
    dt = time.perf_counter() - t0; a1 = capi.alloc_stats()
    print("%-44s wall %9.1f ms  device %9.1f ms  pool allocs +%d frees +%d evict +%d  alloc ms +%.1f" % (
        label, dt * 1e3, capi.last_device_ms(), a1["pool_allocs"] - a0["pool_allocs"], a1["pool_frees"] - a0["pool_frees"],
        a1["evictions"] - a0["evictions"], a1["ms"] - a0["ms"]), flush=True)
    return r

P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
capi.set_device(0)
em4 = EvaluatedMachine.fromMachine(P("psw2dna"), None, useDefaults=True)
dm4 = capi.DeviceMachine(em4)
b4 = capi.DeviceBatch(dm4, *synth_batch(4, 256, 487, 10000, em4.nInTok, em4.nOutTok))
for k in range(2): call("headline materialised Forward, 256 pairs", lambda: b4.forward(capi.MB_MATERIALISE))
del b4
h = HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(20)
em5 = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
dm5 = capi.DeviceMachine(em5)
b5 = capi.DeviceBatch(dm5, *synth_batch(5, 64, 0, 50000, em5.nInTok, em5.nOutTok))
bn = capi.DeviceBatch(dm4, *synth_batch(4, 32, 487, 10000, em4.nInTok, em4.nOutTok))
for rep in range(3):
    call("one-tape rolling Forward 64 x 50 kb (parts)", lambda: b5.forward(capi.MB_ROLLING))
    call("one-tape Viterbi fill 64 x 50 kb (parts)", lambda: b5.viterbi(paths=False))
    if rep == 0: call("one-tape counts 64 x 50 kb", lambda: b5.counts())
    call("psw2dna materialised Forward, 32 pairs", lambda: bn.forward(capi.MB_MATERIALISE))
    call("psw2dna materialised Forward, 32 pairs again", lambda: bn.forward(capi.MB_MATERIALISE))
    call("psw2dna Viterbi with paths, 32 pairs", lambda: bn.viterbi(paths=True))
print(capi.alloc_stats())

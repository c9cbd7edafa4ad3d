"""One-tape sweeps of BASELINE config 5 (fn3 profile, first 20 nodes, composed with simple_introns . translate . dnapsw: 5 063 states)
through the interpreter (MB_WIDE_JIT=0) and through the kernel generated for the machine (mb_wide_jit.cpp): ms per call, and
whether the two agree (Viterbi scores and paths bit for bit, log-likelihoods to 1e-9 relative).
usage: jit_probe.py [nSeq] [length] [nodes]"""
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
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 20
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["roll", "vit", "align", "counts"]
P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
h = HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(nodes)
em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
capi.set_device(0)
res = {}
for jit in ("0", "1"):
    os.environ["MB_WIDE_JIT"] = jit
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch(dm, *synth_batch(5, nSeq, 0, L, em.nInTok, em.nOutTok))
    cells = b.cells()
    out = {}
    def run(label, fn, reps=2):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        out[label] = r
        print("jit=%s %-8s %9.2f ms wall %9.2f ms device  %7.1f G cells/s  %s" % (jit, label, min(ts) * 1e3, capi.last_device_ms(), cells / min(ts) / 1e9, capi.last_kernel_name()), flush=True)
    if "roll" in modes: run("roll", lambda: b.forward(capi.MB_ROLLING))
    if "vit" in modes: run("vit", lambda: b.viterbi(paths=False))
    if "align" in modes: run("align", lambda: b.viterbi())
    if "counts" in modes: run("counts", lambda: b.counts(), 1)
    res[jit] = out
    print(capi.jit_stats(), flush=True)
    del b, dm
a, c = res["0"], res["1"]
if "roll" in a: print("roll   max rel diff %.3g" % float(np.max(np.abs(a["roll"] - c["roll"]) / np.abs(a["roll"]))))
if "vit" in a: print("vit    scores equal:", bool(np.array_equal(a["vit"][0], c["vit"][0])))
if "align" in a: print("align  scores equal: %s  paths equal: %s" % (bool(np.array_equal(a["align"][0], c["align"][0])), bool(np.array_equal(a["align"][2], c["align"][2]))))
if "counts" in a:
    d = np.abs(a["counts"][0] - c["counts"][0]); print("counts max abs diff %.3g rel %.3g; loglike sums %r %r" % (float(d.max()), float((d / np.maximum(np.abs(a["counts"][0]), 1e-300))[a["counts"][0] > 1e-6].max()), a["counts"][1], c["counts"][1]))

"""k workgroups per sequence for the one-tape family: BASELINE config 5's machine, `pairs` sequences of `outlen`
symbols; every mode with one workgroup per sequence (MB_ONETAPE_PARTS=1) and with the parts the library picks (or MB_ONETAPE_PARTS /
MB_ONETAPE_PART_LANES from the command line), results compared: Viterbi scores and paths bit for bit, log-likelihoods and counts to 1e-9.
usage: parts_probe.py [nodes] [pairs] [outlen] [modes: r v p c m] [k,lanes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
outlen = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
modes = sys.argv[4] if len(sys.argv) > 4 else "rvp"
variants = [tuple(int(x) for x in a.split(",")) for a in sys.argv[5:]] or [(0, 0)]      # 0: the library's own choice
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
print("composed %d states %d transitions" % (em.nStates, em.nTransitions), flush=True)
dm = capi.DeviceMachine(em)
inTok, inOff, outTok, outOff = synth_batch(5, pairs, 0, outlen, em.nInTok, em.nOutTok)
b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
cells = b.cells()
def run(mode):
    if mode == "r": return (b.forward(capi.MB_ROLLING),)
    if mode == "m": return (b.forward(capi.MB_MATERIALISE),)
    if mode == "v": return (b.viterbi(paths=False)[0],)
    if mode == "p": r = b.viterbi(); return (r[0], r[1], r[2])
    if mode == "c": r = b.counts(); return (r[0], r[1])
def timed(mode):
    run(mode)
    t0 = time.perf_counter(); r = run(mode); dt = time.perf_counter() - t0
    return r, dt, capi.last_kernel_name()
base = {}
os.environ["MB_ONETAPE_PARTS"] = "1"
for mode in modes:
    base[mode], dt, kn = timed(mode)
    print("one workgroup per sequence  %s  %8.2f G%scells/s  %8.1f ms  %s" % (mode, (2 if mode == "c" else 1) * cells / dt / 1e9, "lattice-" if mode == "c" else "", dt * 1e3, kn), flush=True)
for k, lanes in variants:
    if k: os.environ["MB_ONETAPE_PARTS"] = str(k)
    else: os.environ.pop("MB_ONETAPE_PARTS", None)
    if lanes: os.environ["MB_ONETAPE_PART_LANES"] = str(lanes)
    else: os.environ.pop("MB_ONETAPE_PART_LANES", None)
    for mode in modes:
        r, dt, kn = timed(mode)
        ref = base[mode]
        if mode in "vp": same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(r, ref))
        else: same = max(float(np.max(np.abs(np.asarray(x) - np.asarray(y)) / np.maximum(1e-300, np.abs(np.asarray(y))))) for x, y in zip(r, ref))
        print("parts <= %2d, %4d lanes       %s  %8.2f G%scells/s  %8.1f ms  %s  %s" % (k, lanes, mode, (2 if mode == "c" else 1) * cells / dt / 1e9, "lattice-" if mode == "c" else "", dt * 1e3, kn,
              ("identical" if same is True else "DIFFERENT") if mode in "vp" else "max rel diff %.2e" % same), flush=True)

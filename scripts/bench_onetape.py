"""Config 5: HMMER profile . simple_introns . translate . dnapsw assembled on the box from the fn3
profile truncated to `nodes` nodes (86 = the whole profile: 21 761 states as SURVEY probed it, 20 nodes = 5063 states, the
"~5k states" of the config), composed left to right (MB_COMPOSE_ORDER=right: the boss command line's order), a one-tape generator; Forward (rolling) and Viterbi
fill over `pairs` x `outlen` nt of synthetic DNA, checked against the C oracle on a short prefix."""
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
modes = sys.argv[4] if len(sys.argv) > 4 else "rmv"
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
t = time.perf_counter()
h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
m = (A.composeAll if os.environ.get("MB_COMPOSE_ORDER") == "right" else A.composeLeftToRight)([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
print("composed %d states %d transitions in %.2f s" % (em.nStates, em.nTransitions, time.perf_counter() - t), flush=True)
t = time.perf_counter(); dm = capi.DeviceMachine(em)
inTok, inOff, outTok, outOff = synth_batch(5, pairs, 0, outlen, em.nInTok, em.nOutTok)
b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
cells = b.cells()
def tm(f, name):
    t0 = time.perf_counter(); f(); d0 = time.perf_counter() - t0
    t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print("%-22s %8.2f Gcells/s  %.1f ms (first call %.1f s)  %s" % (name, cells / dt / 1e9, dt * 1e3, d0, capi.last_kernel_name()), flush=True); return r
llr = tm(lambda: b.forward(capi.MB_ROLLING), "forward rolling") if "r" in modes else None
llm = tm(lambda: b.forward(capi.MB_MATERIALISE), "forward materialised") if "m" in modes else None
v = tm(lambda: b.viterbi(paths=False), "viterbi fill") if "v" in modes else None
if "p" in modes:
    r = tm(lambda: b.viterbi(), "viterbi with paths")
    print("  device %.1f ms, %d path edges" % (capi.last_device_ms(), len(r[2])), flush=True)
if "c" in modes:
    b.counts()
    t0 = time.perf_counter(); cnt, lls, _ = b.counts(); dt = time.perf_counter() - t0
    print("fwd+bwd+counts         %8.2f Gcells/s (2 matrices)  %.1f ms (device %.1f)  %s  sum of out-edge counts / symbols = %.6f" %
          (2 * cells / dt / 1e9, dt * 1e3, capi.last_device_ms(), capi.last_kernel_name(), cnt[np.asarray(em.outTok) != 0].sum() / max(1, pairs * outlen)), flush=True)
from oracle import oracle
om = oracle.OracleMachine(em)
n = min(outlen, 300)
short = [(np.zeros(0, np.int32), outTok[outOff[k]:outOff[k] + n]) for k in range(min(pairs, 2))]
bs = capi.DeviceBatch.from_pairs(dm, short)
got = bs.forward(capi.MB_ROLLING)
t0 = time.perf_counter(); ref = [om.loglike(x, y) for x, y in short]; dcpu = time.perf_counter() - t0
print("CPU restatement (1 core, %d nt x %d): %.4f Gcells/s" % (n, len(short), len(short) * (n + 1) * em.nStates / dcpu / 1e9))
exact = [om.loglike(x, y, oracle.SUM_EXACT) for x, y in short]
print("oracle check (%d nt): device %s  exact log-sum-exp %s rel err %.2e;  the reference's table interpolation %s rel err %.2e" % (
    n, got[:2], exact, max(abs(g - r) / abs(r) for g, r in zip(got, exact)), ref, max(abs(g - r) / abs(r) for g, r in zip(got, ref))))
gv = bs.viterbi(paths=False)[0]; rv = [float(om.viterbi(x, y)[-1, -1, -1]) for x, y in short]
print("viterbi bit-exact:", [float(g) for g in gv[:2]] == rv, gv[:2], rv)

"""Retimed one-tape sweep (mb_wide.hip k_wide_retimed) against the levelled column-by-column kernels and the oracle:
Viterbi matrices bit for bit, Forward / Backward matrices within the fast-path tolerance; a few machines and lengths."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from oracle import oracle

P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
def profile(nodes):
    h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
    return EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
gen = Machine.fromJson({"state": [
    {"id": "S", "trans": [{"to": "A"}, {"to": "B", "weight": 0.25}]},
    {"id": "A", "trans": [{"to": "A", "out": "x", "weight": 0.5}, {"to": "B", "out": "y", "weight": 0.3}, {"to": "E", "weight": 0.2}]},
    {"id": "B", "trans": [{"to": "A", "out": "y", "weight": 0.6}, {"to": "B", "out": "x", "weight": 0.1}, {"to": "E", "weight": 0.3}]},
    {"id": "E"}]})
rec = Machine.fromJson(json.loads(json.dumps({"state": [{"id": st.name, "trans": [dict(to=t.dest, weight=t.weight, **({"in": t.out} if t.out else {})) for t in st.trans]} for st in gen.state]})))
cases = [("tiny generator", EvaluatedMachine.fromMachine(gen, {}), 1, (0, 1, 23, 200)),
         ("tiny recogniser", EvaluatedMachine.fromMachine(rec, {}), 0, (0, 1, 23, 200)),
         ("fn3 profile", EvaluatedMachine.fromMachine(HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").machine(True), {}), 1, (0, 1, 40, 300)),
         ("3-node composite", profile(3), 1, (0, 1, 41, 150)),
         ("20-node composite", profile(20), 1, (0, 3, 120))]
if len(sys.argv) > 1 and sys.argv[1] == "whole": cases = [("whole fn3 composite", profile(86), 1, (0, 3, 70))]      # ring in L2
os.environ["MB_WIDE_MIN_STATES"] = "1"; os.environ["MB_WIDE_VITERBI_MIN_STATES"] = "0"      # the one-tape family for every machine and mode
bad = 0
def close(a, b, rel=1e-6, abs_=1e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    with np.errstate(invalid="ignore"):
        return bool(np.all((np.isneginf(a) & np.isneginf(b)) | (np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= abs_ + rel * np.abs(b)))))
for name, em, tape, lens in cases:
    om = oracle.OracleMachine(em)
    nt = em.nOutTok if tape else em.nInTok
    def make(flag):          # the programs are built on first use: every mode once under the flag
        os.environ["MB_WIDE_RETIMED"] = flag
        d = capi.DeviceMachine(em)
        seq = np.ones(2, np.int32); z = np.zeros(0, np.int32)
        bb = capi.DeviceBatch.from_pairs(d, [(z, seq) if tape else (seq, z)] * 2)
        bb.viterbi(paths=False); kv = capi.last_kernel_name(); bb.forward(capi.MB_ROLLING); bb.counts()
        return d, kv
    d0, kv0 = make("0"); d1, kv1 = make("1")
    print(name, "batch kernels:", kv0, kv1)
    for n in lens:
        seq = np.random.RandomState(n).randint(1, nt + 1, size=n).astype(np.int32)
        x, y = (np.zeros(0, np.int32), seq) if tape else (seq, np.zeros(0, np.int32))
        V0 = d0.fill(capi.MB_VITERBI, x, y); k0 = capi.last_kernel_name()
        V1 = d1.fill(capi.MB_VITERBI, x, y); k1 = capi.last_kernel_name()
        F1 = d1.fill(capi.MB_FORWARD, x, y); kf = capi.last_kernel_name()
        B1 = d1.fill(capi.MB_BACKWARD, x, y)
        F0 = d0.fill(capi.MB_FORWARD, x, y); B0 = d0.fill(capi.MB_BACKWARD, x, y)
        okv = np.array_equal(V0, V1)
        oko = np.array_equal(V1, om.viterbi(x, y)) if em.nStates * (n + 1) < 400000 else None
        okf = close(F1, F0) and close(B1, B0)
        b = capi.DeviceBatch.from_pairs(d1, [(x, y)] * 3)
        vll = b.viterbi(paths=False)[0]; ll = b.forward(capi.MB_ROLLING)
        b0 = capi.DeviceBatch.from_pairs(d0, [(x, y)] * 3)
        vll0 = b0.viterbi(paths=False)[0]; ll0 = b0.forward(capi.MB_ROLLING)
        okl = all(v == w for v, w in zip(vll, vll0)) and close(ll, ll0, 1e-9, 1e-12) and vll[0] == V0[-1, -1, -1]
        if not okl: print("   batch:", vll, vll0, V0[-1, -1, -1], V1[-1, -1, -1], ll, ll0, F0[-1, -1, -1], F1[-1, -1, -1])
        print("%-18s S=%5d len %4d  %s vs %s: viterbi identical %s, oracle %s; %s forward/backward close %s; batch %s" % (name, em.nStates, n, k1, k0, okv, oko, kf, okf, okl), flush=True)
        bad += (not okv) + (oko is False) + (not okf) + (not okl)
        if not okv:
            d = np.argwhere(V0 != V1)
            print("   first differing cells:", d[:5].tolist(), V0[tuple(d[0])], V1[tuple(d[0])])
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)

"""CPU tests: the oracle (oracle/mb_oracle.c) against every golden vector the reference's own tests hold for the
DP path (tests/golden/ = data files of /root/reference/t/ and preset/), plus the benchmark-scale anchors
recorded in SURVEY.md section 6.  No GPU needed."""
import json
import math

import numpy as np
import pytest

from conftest import golden_path, load_json, load_matrix_json
from machineboss_amd.seqgen import mt19937_u32, synth_tokens


def tok(em, sp):
    return em.inputTokenizer.tokenize(sp["input"]["sequence"]), em.outputTokenizer.tokenize(sp["output"]["sequence"])


def round_sig(x, n):
    """t/roundfloats.py: round to n significant digits."""
    if x == 0 or not math.isfinite(x):
        return x
    return round(x, n - 1 - int(math.floor(math.log10(abs(x)))))


def test_mt19937_matches_std():
    r = mt19937_u32(5489, 10000)
    assert int(r[0]) == 3499211612 and int(r[9999]) == 4123659995  # C++11 [rand.predef] check value


def test_logsumexp_table_semantics(oracle_mod):
    L = oracle_mod.lib()
    inf = float("inf")
    assert L.mbo_log_sum_exp(-inf, -inf, 0) == -inf            # a == b branch, src/logsumexp.h:79
    assert L.mbo_log_sum_exp(-3.0, -inf, 0) == -3.0
    assert L.mbo_log_sum_exp(0.0, -10.0, 0) == 0.0             # gap >= 10 nats dropped, src/logsumexp.h:53-54
    assert L.mbo_log_sum_exp(0.0, -9.99995, 0) > 0.0
    assert abs(L.mbo_log_sum_exp(1.0, 1.0, 0) - (1.0 + math.log(2))) < 1e-15
    for a, b in [(0.0, -0.12345), (-5.0, -2.5), (3.0, 2.99999)]:
        exact = max(a, b) + math.log1p(math.exp(-abs(a - b)))
        assert abs(L.mbo_log_sum_exp(a, b, 0) - exact) < 1e-9   # interpolation error bound h^2/8*f''
        assert abs(L.mbo_log_sum_exp(a, b, 1) - exact) < 1e-15


def test_forward_matrix_golden(oracle_mod, machines):
    """test-fwd-bitnoise-params-tiny (Makefile:493): full Forward matrix at setprecision(5)."""
    m, em = machines("bitnoise", load_json("io", "params.json"))
    i, o = tok(em, load_json("io", "tiny.json"))
    F = oracle_mod.OracleMachine(em).forward(i, o)
    exp = load_matrix_json("expect", "fwd-bitnoise-params-tiny.json")
    assert len(exp) == 16
    for (ip, op, _), v in exp.items():
        got = F[op, ip, 0]
        assert (got == v) if not math.isfinite(v) else float("%.5g" % got) == v


def test_backward_matrix_golden(oracle_mod, machines):
    """test-back-bitnoise-params-tiny (Makefile:496)."""
    m, em = machines("bitnoise", load_json("io", "params.json"))
    i, o = tok(em, load_json("io", "tiny.json"))
    B = oracle_mod.OracleMachine(em).backward(i, o)
    exp = load_matrix_json("expect", "back-bitnoise-params-tiny.json")
    assert len(exp) == 16
    for (ip, op, _), v in exp.items():
        got = B[op, ip, 0]
        assert (got == v) if not math.isfinite(v) else float("%.5g" % got) == v


def test_counts_golden(oracle_mod, machines):
    """test-fb-bitnoise-params-tiny (Makefile:499): [[1,1,1,0]]."""
    m, em = machines("bitnoise", load_json("io", "params.json"))
    i, o = tok(em, load_json("io", "tiny.json"))
    c = np.zeros(em.nTransitions)
    ll = oracle_mod.OracleMachine(em).counts_add(i, o, c)
    expect = load_json("expect", "fwdback-bitnoise-params-tiny.json")
    assert [float("%.6g" % x) for x in c] == [float(x) for x in expect[0]]
    assert float("%.5g" % ll) == -4.6253


def test_viterbi_alignment_golden(oracle_mod, machines):
    """test-align-stutter-noise (Makefile:515): Viterbi traceback through silent states, meta.path."""
    m, em = machines("bitstutter-noise", load_json("io", "params.json"))
    sp = load_json("io", "difflen.json")[0]
    i, o = tok(em, sp)
    om = oracle_mod.OracleMachine(em)
    V = om.viterbi(i, o)
    path = om.traceback(i, o, V)
    exp = load_json("expect", "align-stutter-noise-difflen.json")[0]["meta"]["path"]
    assert exp["start"] == 0
    got = []
    for e in path:
        t = m.state[int(em.src[e])].getTransition(int(em.transIndex[e]))
        d = {"to": t.dest}
        if t.inp: d["in"] = t.inp
        if t.out: d["out"] = t.out
        got.append(d)
    want = [{k: v for k, v in tr.items() if k in ("to", "in", "out")} for tr in exp["trans"]]
    assert got == want
    # the alignment columns implied by the path (src/seqpair.cpp:83-89)
    cols = [[d.get("in", ""), d.get("out", "")] for d in got if "in" in d or "out" in d]
    assert cols == load_json("expect", "align-stutter-noise-difflen.json")[0]["alignment"]


def test_loglike_goldens(oracle_mod, machines):
    """test-101-bitnoise-001, test-101-bitstutternoise-0011 (Makefile:567-572), rounded as t/roundfloats.py does."""
    p = load_json("io", "params.json")
    _, em = machines("bitnoise", p)
    om = oracle_mod.OracleMachine(em)
    i = em.inputTokenizer.tokenize(load_json("io", "seq101.json")["sequence"])
    o = em.outputTokenizer.tokenize(load_json("io", "seq001.json")["sequence"])
    assert round_sig(om.loglike(i, o), 4) == load_json("expect", "101-bitnoise-001.json")[0][0]
    _, em2 = machines("bitstutter-noise", p)
    om2 = oracle_mod.OracleMachine(em2)
    i = em2.inputTokenizer.tokenize(list("101")); o = em2.outputTokenizer.tokenize(list("0011"))
    assert round_sig(om2.loglike(i, o), 3) == load_json("expect", "101-bitstutternoise-fwd-0011.json")[0][0]
    assert round_sig(om2.viterbi(i, o)[-1, -1, -1], 3) == load_json("expect", "101-bitstutternoise-vit-0011.json")[0][0]
    # rolling and full matrices agree (quirk Q8)
    assert om2.loglike(i, o) == om2.forward(i, o)[-1, -1, -1]


def test_param_counts_goldens(oracle_mod, machines):
    """test-counts2 (Makefile:521): {"p":2,"q":1}; test-counts3 (Makefile:524): {"p":3}."""
    from machineboss_amd.dp import MachineCounts
    p = load_json("io", "params.json")
    m, em = machines("bitnoise", p)
    c = np.zeros(em.nTransitions)
    oracle_mod.OracleMachine(em).counts_add(em.inputTokenizer.tokenize(list("101")), em.outputTokenizer.tokenize(list("001")), c)
    mc = MachineCounts.__new__(MachineCounts); mc.machine = em; mc._flat = c; mc.loglike = 0.0
    pc = mc.paramCounts(m, p)
    assert {k: float("%.6g" % v) for k, v in pc.items()} == load_json("expect", "counts.json")
    m2, em2 = machines("counter", {})
    c2 = np.zeros(em2.nTransitions)
    oracle_mod.OracleMachine(em2).counts_add(np.zeros(0, np.int32), em2.outputTokenizer.tokenize(list("xxx")), c2)
    mc2 = MachineCounts.__new__(MachineCounts); mc2.machine = em2; mc2._flat = c2; mc2.loglike = 0.0
    # counter.json defines p through "defs": the count is still reported against p
    m2.funcs = {}
    assert {k: float("%.6g" % v) for k, v in mc2.paramCounts(m2, {"p": 1}).items()} == load_json("expect", "counter.json")


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_survey_anchors(oracle_mod, machines, idx):
    """Benchmark-scale reference outputs recorded in SURVEY.md section 6 (see survey_anchors.json provenance)."""
    a = load_json("survey_anchors.json")["anchors"][idx]
    m, em = machines(a["preset"], None, useDefaults=True, preset=True)
    assert (em.nStates, em.nTransitions) == (a["states"], a["trans"])
    om = oracle_mod.OracleMachine(em)
    i, o = synth_tokens(a["seed"], a["inLen"], a["outLen"], em.nInTok, em.nOutTok)
    assert float("%.10g" % om.loglike(i, o)) == a["forward"]
    if a["forward_exact"] is not None:
        assert float("%.10g" % om.loglike(i, o, oracle_mod.SUM_EXACT)) == a["forward_exact"]
    V = om.viterbi(i, o)
    assert float("%.10g" % V[-1, -1, -1]) == a["viterbi"]
    assert len(om.traceback(i, o, V)) == a["pathLen"]


def test_iteration_orders(oracle_mod, machines):
    """The CSR orders equal the nested-map iteration orders (src/eval.h:66-68); appendix A of SURVEY.md."""
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em)
    assert list(om.incoming_order()) == list(em.incomingOrder())
    assert list(om.outgoing_order()) == list(em.outgoingOrder())
    inc = [(int(em.dst[e]), int(em.inTok[e]), int(em.outTok[e]), int(em.src[e]), int(em.transIndex[e])) for e in om.incoming_order()]
    # dst0 S : 16 match edges from M (state 5), ti 1..16; dst2 W: silent from S ti1, J ti1
    assert [x for x in inc if x[0] == 0] == [(0, a, b, 5, 1 + 4 * (a - 1) + (b - 1)) for a in range(1, 5) for b in range(1, 5)]
    assert [x for x in inc if x[0] == 2] == [(2, 0, 0, 0, 1), (2, 0, 0, 1, 1)]
    assert list(em.transOffset) == [0, 2, 4, 6, 8, 12, 29, 34, 34]
    assert list(em.silentLevels()) == [0, 0, 1, 0, 1, 2, 2, 3]
    assert abs(em.logWeight[0] - math.log(0.5)) == 0


def test_empty_sequences(oracle_mod, machines):
    """1x1 lattice = pure silent propagation (Machine::downsample's use, src/machine.cpp:2053-2076)."""
    m, em = machines("silent", {"u": .5, "v": .5, "w": .5, "x": .5, "y": .5, "z": .5})
    om = oracle_mod.OracleMachine(em)
    e = np.zeros(0, np.int32)
    F = om.forward(e, e)
    assert F.shape == (1, 1, 7) and F[0, 0, 0] == 0 and F[0, 0, 1] == math.log(.5) and F[0, 0, 2] == -math.inf
    i = em.inputTokenizer.tokenize(["A"]); o = em.outputTokenizer.tokenize(["B"])
    assert abs(om.loglike(i, o) - 6 * math.log(.5)) < 1e-12
    B = om.backward(i, o)
    assert abs(B[0, 0, 0] - 6 * math.log(.5)) < 1e-12


# ---- outputs of the reference's own JavaScript CPU tier, run in the dev container (tests/golden/make_js_goldens.mjs) ----
def _js_cases():
    cases = {c["name"]: c for c in load_json("js", "cases.json")}
    return [(cases[g["name"]], g) for g in load_json("js", "goldens.json")]


def _num(v):
    return -math.inf if v == "-inf" else (math.inf if v == "inf" else float(v))


@pytest.mark.parametrize("idx", range(10))
def test_oracle_matches_reference_js_tier(oracle_mod, idx):
    """The reference ships a second, independent implementation of this DP path (js/webgpu/cpu/*-2d.mjs: Float64, exact
    logsumexp, dense transition tensor).  Its outputs on five machines -- generated by running THAT code here with node --
    pin the oracle: Forward/Backward log-likelihoods and every cell to 1e-10 relative, Viterbi scores bit for bit.
    Cases 5-9 (round 6): ONE-TAPE machines through the reference's 1-D tier (js/webgpu/cpu/{forward,backward,viterbi}-1d.mjs) -- fn3
    profile truncations composed with `translate` (generators of DNA; one turned into a recogniser), a Plan7-flanked profile, and the
    2-node form of BASELINE config 5's composition -- the family the oracle otherwise covers only through its 2-D restatement with
    inLen = 0."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    case, gold = _js_cases()[idx]
    m = Machine.fromFile(golden_path(*case["machine"].split("/")))
    defs = m.getParamDefs(True); defs.update(case["params"])
    em = EvaluatedMachine.fromMachine(m, defs)
    om = oracle_mod.OracleMachine(em)
    x = np.array(case["input"], np.int32); y = np.array(case["output"], np.int32)
    F = om.forward(x, y, oracle_mod.SUM_EXACT); B = om.backward(x, y, oracle_mod.SUM_EXACT); V = om.viterbi(x, y)
    rel = lambda a, b: abs(a - b) <= 1e-10 * max(1.0, abs(b))
    assert rel(F[-1, -1, -1], _num(gold["forward"])) and rel(B[0, 0, 0], _num(gold["backward"]))
    assert V[-1, -1, -1] == _num(gold["viterbi"])
    assert rel(om.loglike(x, y, oracle_mod.SUM_TABLE), _num(gold["forward"])) or abs(om.loglike(x, y) - _num(gold["forward"])) < 1e-4
    S = em.nStates; Lo = len(y)

    def cell(M, k):   # JS layout [(i*(Lo+1)+o)*S+s] -> ours [o][i][s]; the 1-D tier: [p*S+s], p along the machine's one tape
        s = k % S; io = k // S
        if gold.get("oneTape") == "in": return M[0, io, s]
        return M[io % (Lo + 1), io // (Lo + 1), s]
    if "forwardCells" in gold:
        pairs = [(k, _num(f), _num(b)) for k, (f, b) in enumerate(zip(gold["forwardCells"], gold["backwardCells"]))]
    else:
        pairs = [(k, _num(f), _num(b)) for k, f, b in gold["sample"]]
    assert len(pairs) > 10
    for k, f, b in pairs:
        for got, ref in ((cell(F, k), f), (cell(B, k), b)):
            assert (got == ref) if not math.isfinite(ref) else rel(got, ref), (k, got, ref)

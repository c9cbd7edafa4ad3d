"""CPU tests of transducer composition (machineboss_amd/algebra.py) against the reference's expected machines
(Makefile:247-270 COMPOSE_TESTS: `boss A.json B.json` prints the composition) and its probe numbers."""
import json

import pytest

from conftest import golden_path, load_json
from machineboss_amd import algebra as A
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.machine import Constraints, Machine, MachineError


def _norm(j):
    """An expected machine file, normalised like algebra.machineToJson (state numbers, default weights dropped)."""
    out = {"state": []}
    for n, s in enumerate(j["state"]):
        sj = {"n": n}
        if "id" in s:
            sj["id"] = s["id"]
        tr = []
        for t in s.get("trans", []):
            tj = {"to": t["to"]}
            if t.get("in"): tj["in"] = t["in"]
            if t.get("out"): tj["out"] = t["out"]
            if "weight" in t and not A.wIsOne(t["weight"]): tj["weight"] = A.weightToJson(A._canon(t["weight"]))
            tr.append(tj)
        if tr:
            sj["trans"] = tr
        out["state"].append(sj)
    for k in ("defs", "cons"):
        if k in j:
            out[k] = j[k]
    return out


M = lambda n: Machine.fromFile(golden_path("machine", n + ".json"))
P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))


@pytest.mark.parametrize("a,b,expect,showParams", [
    ("bitecho", "bitecho", "bitecho-bitecho", False), ("bitecho", "bitstutter", "bitecho-bitstutter", False),
    ("bitstutter", "bitstutter", "bitstutter-bitstutter", False), ("bitnoise", "bitnoise", "bitnoise-bitnoise", True),
    ("unitindel", "unitindel", "unitindel-unitindel", True)])
def test_compose_reference_goldens(a, b, expect, showParams):
    """State order (advanceSort, incl. the null-padding retry), state names, symbolic weights (sums of products with the
    reference's simplifications) and merged defs/cons are exactly the reference's."""
    got = A.machineToJson(A.compose(M(a), M(b)), showParams)
    assert got == _norm(load_json("expect", expect + ".json"))


def test_compose_matches_shipped_composition():
    """t/machine/bitstutter-noise.json is bitstutter composed with bitnoise (test-align-stutter-noise, Makefile:515)."""
    assert A.machineToJson(A.compose(M("bitstutter"), M("bitnoise"))) == A.machineToJson(M("bitstutter-noise"))


def test_compose_presets_probe_numbers():
    """protpsw.translate = 177 states / 1502 transitions / 5 silent levels; the literal three-way
    composition aborts on shared parameter names; with dnapsw's constraints cleared it has 482 states / 3095 transitions."""
    pt = A.compose(P("protpsw"), P("translate"))
    em = EvaluatedMachine.fromMachine(pt, None, useDefaults=True)
    assert (em.nStates, em.nTransitions, int(em.silentLevels().max()) + 1) == (177, 1502, 5)
    with pytest.raises(MachineError, match="Inconsistent constraints for eqmA"):
        A.compose(pt, P("dnapsw"))
    d = P("dnapsw"); d.cons = Constraints()
    ptd = A.compose(pt, d)
    assert (len(ptd.state), ptd.nTransitions()) == (482, 3095) and ptd.isAdvancingMachine()
    assert A.composeAll([P("protpsw"), P("translate")]).nTransitions() == 1502


def test_weight_algebra_simplifications():
    """src/weight.cpp:137-182."""
    assert A.wMultiply(1, "p") == "p" and A.wMultiply("p", 1.0) == "p" and A.wMultiply(0, "p") == 0
    assert A.wMultiply(2, 3) == 6 and A.wMultiply(0.5, 4) == 2.0 and A.wMultiply("p", "q") == {"*": ["p", "q"]}
    assert A.wAdd(0, "p") == "p" and A.wAdd(1, 2) == 3 and A.wAdd("p", {"-": [0, "q"]}) == {"-": ["p", "q"]}
    assert A.weightToJson(A.wGeometricSum("p")) == {"geomsum": "p"} and A.weightToJson(A.wNegate("p")) == {"not": "p"}
    assert A._canon({"not": "p"}) == {"-": [1, "p"]}


def test_silent_cycle_summation():
    """advancingMachine (src/machine.cpp:1177-1230): a silent back-transition is summed out as a geometric series."""
    m = Machine.fromJson({"state": [{"id": "a", "trans": [{"to": "b"}]},
                                     {"id": "b", "trans": [{"to": "a", "weight": "r"}, {"to": "c", "out": "x", "weight": "s"}]},
                                     {"id": "c"}]})
    assert not m.isAdvancingMachine()
    am = A.advancingMachine(m)
    assert am.isAdvancingMachine() and len(am.state) == 3
    # state b: b -> a -> b is a silent self-loop of weight r: exits are scaled by 1/(1-r)
    w = [t.weight for t in am.state[1].trans if t.out == "x"][0]
    assert A.weightToJson(A._canon(w)) == {"*": [{"geomsum": "r"}, "s"]}


# ---- HMMER3 profile importer (src/hmmer.cpp) against the reference's construct-test goldens ------------------------------
def _round3(x):
    return float("%.3g" % x)


@pytest.mark.parametrize("golden,build", [("fn3.json", lambda h: h.machine(False)), ("fn3-plan7.json", lambda h: h.plan7Machine(False)),
                                          ("fn3-multihit.json", lambda h: h.plan7Machine(True))])
def test_hmmer_importer_goldens(golden, build):
    """`boss --hmmer-global / --hmmer-plan7 / --hmmer-multihit t/hmmer/fn3.hmm` (Makefile test-hmmer*, 3 significant digits)."""
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd.algebra import machineToJson
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm"))
    assert len(h.node) == 86 and len(h.alph) == 20
    got = machineToJson(build(h)); want = load_json("expect", golden)
    assert len(got["state"]) == len(want["state"])
    for a, b in zip(got["state"], want["state"]):
        assert a.get("id") == b.get("id") and len(a.get("trans", [])) == len(b.get("trans", []))
        for ta, tb in zip(a.get("trans", []), b.get("trans", [])):
            assert ta["to"] == tb["to"] and ta.get("out") == tb.get("out") and ta.get("in") == tb.get("in")
            assert _round3(ta.get("weight", 1)) == pytest.approx(_round3(tb.get("weight", 1)), rel=1e-2)


def test_hmmer_local_mode_and_profile_composition():
    """Local mode: entry weights occ[k]/Z (p7_ProfileConfig), unit exits from every match / delete state; and BASELINE
    config 5's machine at test size -- profile . simple_introns . translate . dnapsw is a one-tape advancing machine."""
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd import algebra as A
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm"))
    m = h.machine(True)
    assert len(m.state) == 5 * 86 + 4 and len(m.state[0].trans) == 85
    assert abs(sum(t.weight * (87 - k) for k, t in enumerate(m.state[0].trans, 1)) - 1.0) < 1e-12
    assert [t.dest for t in m.state[h.d_idx(3)].trans][-1] == h.core_end_idx() and m.state[h.d_idx(3)].trans[-1].weight == 1
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    c = A.composeAll([h.truncated(3).machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    em = EvaluatedMachine.fromMachine(c, None, useDefaults=True)
    assert (em.nStates, em.nTransitions, em.nInTok, em.nOutTok) == (600, 1989, 0, 4)
    # SURVEY.md section 8(d) probed the reference composing from the left; at 5 nodes the two orders give 1268 / 998 states
    five = [h.truncated(5).machine(True), P("simple_introns"), P("translate"), P("dnapsw")]
    assert len(A.composeLeftToRight(five).state) == 1268 and len(A.composeAll(five).state) == 998


def test_config5_machine_matches_survey_probe():
    """The whole fn3 profile . simple_introns . translate . dnapsw, composed pairwise from the left like the probe of the
    reference in SURVEY.md section 8(d) row 5: 21 761 states, 63 267 transitions (32 340 out-only, 30 927 silent), 1554
    silent levels, no input alphabet.  Pins the HMMER importer and the composition together (about a minute of host time)."""
    from machineboss_amd.hmmer import HmmerModel
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm"))
    c = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    em = EvaluatedMachine.fromMachine(c, None, useDefaults=True)
    outOnly = sum(1 for i, o in zip(em.inTok, em.outTok) if i == 0 and o != 0)
    silent = sum(1 for i, o in zip(em.inTok, em.outTok) if i == 0 and o == 0)
    assert (em.nStates, em.nTransitions, outOnly, silent, em.nInTok) == (21761, 63267, 32340, 30927, 0)
    lev = [0] * em.nStates
    for s_, d_, i, o in zip(em.src, em.dst, em.inTok, em.outTok):      # edges are in ascending source order
        if i == 0 and o == 0 and s_ < d_:
            lev[d_] = max(lev[d_], lev[s_] + 1)
    assert max(lev) + 1 == 1554

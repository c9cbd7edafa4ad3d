"""Step 1 of the JS-tier goldens: write tests/golden/js/cases.json (machine file, numeric parameters, token sequences).

The cases are then run through the REFERENCE's own JavaScript CPU implementation by make_js_goldens.mjs (step 2).
Run from the repo root:  python tests/golden/make_js_cases.py && node tests/golden/make_js_goldens.mjs
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens

G = os.path.join(ROOT, "tests", "golden")
io_params = json.load(open(os.path.join(G, "io", "params.json")))
cases = []


def add(name, rel, params, useDefaults, seqs=None, seed=None, il=0, ol=0):
    m = Machine.fromFile(os.path.join(G, rel))
    em = EvaluatedMachine.fromMachine(m, params, useDefaults=useDefaults)
    defs = m.getParamDefs(useDefaults)
    if params:
        defs.update(params)
    numeric = {k: v for k, v in defs.items() if isinstance(v, (int, float))}
    if seqs is not None:
        x = em.inputTokenizer.tokenize(list(seqs[0])); y = em.outputTokenizer.tokenize(list(seqs[1]))
    else:
        x, y = synth_tokens(seed, il, ol, em.nInTok, em.nOutTok)
    cases.append({"name": name, "machine": rel, "params": numeric, "input": [int(t) for t in x], "output": [int(t) for t in y],
                  "nStates": em.nStates})


add("bitnoise-001-101", "machine/bitnoise.json", io_params, False, seqs=("001", "101"))
add("bitstutter-noise-101-10011", "machine/bitstutter-noise.json", io_params, False, seqs=("101", "10011"))
add("dnapsw-12x15", "preset/dnapsw.json", None, True, seed=5, il=12, ol=15)
add("protpsw-9x11", "preset/protpsw.json", None, True, seed=6, il=9, ol=11)
add("psw2dna-4x13", "preset/psw2dna.json", None, True, seed=7, il=4, ol=13)

# ---- one-tape machines for the reference's 1-D tier (js/webgpu/cpu/{forward,backward,viterbi}-1d.mjs): generators assembled by this
# repository's importer and composition (hmmer.py, algebra.py; both pinned to the reference's own expected machines elsewhere), written
# out as plain numeric machines -- `weight` = exp(log weight of the evaluated machine) -- so that the reference's JavaScript and the
# tests read the same doubles.  The dense tier is O(L S^2): a few hundred states.
import math
import numpy as np
from machineboss_amd import algebra as A
from machineboss_amd.hmmer import HmmerModel


def write_numeric(em, rel, swap=False):
    """the evaluated machine as Machine Boss JSON with numeric weights; swap: outputs become inputs (a recogniser)"""
    inA = [""] + list(em.inputTokenizer.tok2sym[1:]) if em.nInTok else [""]
    outA = [""] + list(em.outputTokenizer.tok2sym[1:]) if em.nOutTok else [""]
    states = [{"n": s, "trans": []} for s in range(em.nStates)]
    for e in range(em.nTransitions):
        t = {"to": int(em.dst[e])}
        i, o = inA[int(em.inTok[e])], outA[int(em.outTok[e])]
        if swap: i, o = o, i
        if i: t["in"] = i
        if o: t["out"] = o
        t["weight"] = math.exp(float(em.logWeight[e]))
        states[int(em.src[e])]["trans"].append(t)
    os.makedirs(os.path.dirname(os.path.join(G, rel)), exist_ok=True)
    json.dump({"state": states}, open(os.path.join(G, rel), "w"))


def add1d(name, machine, L, seed, swap=False, ntok=None):
    em0 = EvaluatedMachine.fromMachine(machine, None, useDefaults=True)
    rel = "js/machines/%s.json" % name
    write_numeric(em0, rel, swap)
    em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(G, rel)), {})
    assert em.nStates == em0.nStates and em.nTransitions == em0.nTransitions
    nt = ntok or (em.nInTok if swap else em.nOutTok)      # (ntok = 3: DNA over {A, C, G} -- no stop codons, which `translate` does not emit)
    seq = [int(t) for t in np.random.RandomState(seed).randint(1, nt + 1, size=L)]
    cases.append({"name": name, "machine": rel, "params": {}, "input": seq if swap else [], "output": [] if swap else seq, "nStates": em.nStates,
                  "oneTape": "in" if swap else "out"})


P = lambda n: Machine.fromFile(os.path.join(G, "preset", n + ".json"))
h = HmmerModel.fromFile(os.path.join(G, "hmmer", "fn3.hmm"))
add1d("fn3-3-translate", A.composeLeftToRight([h.truncated(3).machine(True), P("translate")]), 42, 11, ntok=3)
add1d("fn3-6-translate", A.composeLeftToRight([h.truncated(6).machine(True), P("translate")]), 120, 12, ntok=3)
add1d("fn3-4-plan7", A.advancingMachine(h.truncated(4).plan7Machine(False)) if not h.truncated(4).plan7Machine(False).isAdvancingMachine() else h.truncated(4).plan7Machine(False), 60, 13)
add1d("fn3-3-translate-recogniser", A.composeLeftToRight([h.truncated(3).machine(True), P("translate")]), 51, 14, swap=True, ntok=3)
add1d("fn3-2-introns-translate-dnapsw", A.composeLeftToRight([h.truncated(2).machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), 200, 15)
json.dump(cases, open(os.path.join(G, "js", "cases.json"), "w"), indent=1)
print("wrote", len(cases), "cases")

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
json.dump(cases, open(os.path.join(G, "js", "cases.json"), "w"), indent=1)
print("wrote", len(cases), "cases")

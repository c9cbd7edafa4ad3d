"""CPU tests of the host logic and of the C-ABI library's export surface (no compute calls)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_path, load_json
from machineboss_amd.machine import Machine, MachineError, evalWeight
from machineboss_amd.evalmachine import EvaluatedMachine, Tokenizer


def test_weight_expressions():
    d = {"p": 0.25, "q": {"not": "p"}, "r": {"*": ["p", "q"]}}
    assert evalWeight("q", d) == 0.75
    assert evalWeight("r", d) == 0.25 * 0.75
    assert evalWeight({"/": [1, {"-": [1, "p"]}]}, d) == 1 / 0.75
    assert evalWeight({"geomsum": "p"}, d) == 1 / 0.75
    assert evalWeight({"exp": {"log": "p"}}, d) == math.exp(math.log(0.25))
    assert evalWeight(True, d) == 1.0 and evalWeight(None, d) == 0.0
    with pytest.raises(MachineError):
        evalWeight("zz", d)
    with pytest.raises(MachineError):
        evalWeight({"p2": "p"}, d)
    with pytest.raises(MachineError):  # cyclic definition
        evalWeight("a", {"a": {"not": "a"}})


def test_default_params_and_alphabets():
    m = Machine.fromFile(golden_path("preset", "dnapsw.json"))
    p = m.getParamDefs(True)
    assert p["gapOpen"] == 0.5 and p["eqmA"] == 0.25 and p["subAC"] == 0.25
    assert m.inputAlphabet() == ["A", "C", "G", "T"] and m.outputAlphabet() == ["A", "C", "G", "T"]
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    assert em.nStates == 8 and em.nTransitions == 34
    # parameter-free constructor: all log-weights zero (src/eval.cpp:59)
    em0 = EvaluatedMachine.fromMachine(m)
    assert not em0.logWeight.any()
    with pytest.raises(MachineError):
        EvaluatedMachine.fromMachine(m, {})  # parameters missing


def test_state_references_and_errors():
    j = {"state": [{"id": "a", "trans": [{"to": "b", "in": "x"}]}, {"id": "b"}]}
    m = Machine.fromJson(j)
    assert m.state[0].trans[0].dest == 1
    with pytest.raises(MachineError):
        Machine.fromJson({"state": [{"id": "a", "trans": [{"to": "zz"}]}]})
    with pytest.raises(MachineError):
        Machine.fromJson({"state": [{"n": 1}]})
    # not advancing: silent edge backwards from a state >= 1 (src/machine.cpp:758-764)
    bad = Machine.fromJson({"state": [{"trans": [{"to": 1}]}, {"trans": [{"to": 1}]}]})
    assert not bad.isAdvancingMachine()
    with pytest.raises(MachineError):
        EvaluatedMachine.fromMachine(bad, {})


def test_tokenizer():
    t = Tokenizer(["A", "C"])
    assert list(t.tokenize(["C", "A"])) == [2, 1] and t.detokenize([1, 2]) == ["A", "C"]
    assert not t.canTokenize(["G"])
    with pytest.raises(MachineError):
        t.tokenize(["G"])


def test_abi_exports_match_header():
    """The shared library loads and exports every symbol include/mbhip.h declares."""
    from machineboss_amd import build, capi
    build.build()
    hdr = open(os.path.join(ROOT, "include", "mbhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mb_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared


def test_no_cpu_fallback_without_gpu():
    """Product path must fail loudly when there is no device (this test only asserts on GPU-less hosts)."""
    from machineboss_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    m = Machine.fromFile(golden_path("machine", "bitnoise.json"))
    em = EvaluatedMachine.fromMachine(m, load_json("io", "params.json"))
    with pytest.raises(capi.MbError):
        capi.DeviceMachine(em)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing in the package may import, link or dlopen it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|libmboracle|\bmbo_[a-z]|mb_oracle\.c", re.M)
    pkg = os.path.join(ROOT, "machineboss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), (dirpath, f)


# ---- Envelope / SeqPair (src/seqpair.{h,cpp}); expectations are the reference's own (Makefile:450-462 test-env) ----------
ENV_CASES = [("tinypath", "full", "tinypath_full_env"), ("tinypath", "path", "tinypath_path_env"),
             ("smallpath", "path", "smallpath_path_env"), ("smallpath", 0, "smallpath_area0_env"),
             ("smallpath", 1, "smallpath_area1_env"), ("smallpath", 2, "smallpath_area2_env"),
             ("smallpath", 3, "smallpath_area3_env"), ("smallpath", 4, "smallpath_area4_env"),
             ("smallpath", 5, "smallpath_area4_env"), ("asympath", 0, "asympath_area0_env"),
             ("asympath", 1, "asympath_area1_env"), ("asympath", "path", "asympath_area0_env")]


@pytest.mark.parametrize("io,kind,expect", ENV_CASES)
def test_envelope_reference_goldens(io, kind, expect):
    """t/src/testenv.cpp: initFull / initPath / initPathArea(width) print exactly the reference's expected envelopes."""
    from machineboss_amd.seqpair import Envelope, SeqPair
    sp = SeqPair.fromJson(load_json("io", io + ".json"))
    env = Envelope()
    if kind == "full":
        env.initFull(sp)
    elif kind == "path":
        env.initPath(sp.alignment)
    else:
        env.initPathArea(sp.alignment, kind)
    assert env.writeJson() == open(golden_path("expect", expect + ".json")).read().strip()
    assert env.fits(sp) and env.connected()
    off = env.offsets()
    assert off[0] == 0 and off[-1] == sum(e - s for s, e in zip(env.inStart, env.inEnd))


def test_seqpair_alignment_defaults():
    """SeqPair::readJson (src/seqpair.cpp:8-38): sequences default to the alignment's columns; Envelope(sp) picks the path."""
    from machineboss_amd.seqpair import Envelope, SeqPair, seqPairListFromJson
    pl = seqPairListFromJson(load_json("io", "pathlist.json"))
    assert pl[0].input == ["0", "0", "1"] and pl[0].output == ["1", "0", "1"] and pl[0].inputName == "001"
    e = Envelope(pl[0])
    assert not e.isFull() and e.writeJson() == "[[0,1],[1,2],[2,3],[3,4]]"
    assert Envelope(seqPairListFromJson(load_json("io", "seqpairlist.json"))[0]).isFull()
    assert Envelope(pl[0], 1).writeJson() == Envelope.pathAreaEnvelope(pl[0].alignment, 1).writeJson()
    bad = Envelope(); bad.inLen, bad.outLen, bad.inStart, bad.inEnd = 3, 1, [0, 3], [1, 4]
    assert not bad.connected()


def test_oracle_envelope_semantics(oracle_mod):
    """Oracle fills under an envelope: cells outside are -inf, a full envelope changes nothing, and a path envelope of a
    one-path machine leaves exactly that path's likelihood."""
    from machineboss_amd.seqpair import Envelope, SeqPair
    m = Machine.fromFile(golden_path("machine", "bitnoise.json"))
    em = EvaluatedMachine.fromMachine(m, load_json("io", "params.json"))
    om = oracle_mod.OracleMachine(em)
    sp = SeqPair.fromJson(load_json("io", "tinypath.json"))
    x, y = em.inputTokenizer.tokenize(sp.input), em.outputTokenizer.tokenize(sp.output)
    full = om.forward(x, y, oracle_mod.SUM_EXACT)
    env = Envelope(sp)
    with oracle_mod.envelope(env.inStart, env.inEnd):
        F = om.forward(x, y, oracle_mod.SUM_EXACT); B = om.backward(x, y, oracle_mod.SUM_EXACT)
        c = np.zeros(em.nTransitions); ll = om.counts_add(x, y, c, oracle_mod.SUM_EXACT)
    for o in range(len(y) + 1):
        for i in range(len(x) + 1):
            if not env.contains(i, o):
                assert np.all(np.isneginf(F[o, i])) and np.all(np.isneginf(B[o, i]))
    # bitnoise is a 1-state machine: the path envelope admits exactly one path (3 match steps)
    assert F[-1, -1, -1] <= full[-1, -1, -1] and abs(B[0, 0, 0] - F[-1, -1, -1]) < 1e-12 and ll == F[-1, -1, -1]
    assert abs(c.sum() - 3.0) < 1e-9
    fe = Envelope.fullEnvelope(sp)
    with oracle_mod.envelope(fe.inStart, fe.inEnd):
        assert np.array_equal(om.forward(x, y, oracle_mod.SUM_EXACT), full)


def test_host_logsumexp_helpers_match_reference_table(oracle_mod):
    """mb_log_sum_exp / _n / mb_log_inner_product (src/logsumexp.h:72-172: the helpers that stay exported when logsumexp.*
    is replaced): the reference's interpolated-table arithmetic bit for bit -- against the oracle's restatement, which is
    pinned to the reference's golden matrices -- incl. -inf handling and the 10-nat cut-off."""
    import ctypes as C
    from machineboss_amd import capi
    L = capi.load(); O = oracle_mod.lib()
    rng = np.random.RandomState(5)
    inf = float("inf")
    pairs = [(-inf, -inf), (-inf, 1.5), (1.5, -inf), (0.0, 0.0), (1.0, -9.0), (1.0, -9.00001), (3.0, 3.0 + 1e-9), (2.0, -8.0)]
    pairs += [tuple(rng.uniform(-40, 5, 2)) for _ in range(20000)]
    for a, b in pairs:
        assert L.mb_log_sum_exp(a, b) == O.mbo_log_sum_exp(a, b, 0), (a, b)
    assert L.mb_log_sum_exp(1.0, -9.5) == 1.0                 # terms 10 nats below the maximum are dropped entirely
    v = rng.uniform(-12, 0, 37); w = rng.uniform(-3, 0, 37)
    tot = -inf; lip = -inf
    for x, y in zip(v, w):
        tot = O.mbo_log_sum_exp(tot, x, 0); lip = O.mbo_log_sum_exp(lip, x + y, 0)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert L.mb_log_sum_exp_n(p(v), len(v)) == tot and L.mb_log_inner_product(p(v), p(w), None, len(v)) == lip
    assert L.mb_log_sum_exp_n(p(v), 0) == -inf


def test_logsumexp_header_compiles(tmp_path):
    """machineboss_amd/cxx/mb_logsumexp.hpp: the reference's helper names over the C-ABI."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "mb_logsumexp.hpp"\n#include <cstdio>\nusing namespace MachineBossHIP;\n'
                   'int main() { double a = -1; log_accum_exp(a, -2); std::vector<double> v{-1, -2, -3};\n'
                   '  std::printf("%.17g %.17g %.17g\\n", a, log_sum_exp(v), logInnerProduct(v, v)); return log_sum_exp(-1., -2., -3.) == log_sum_exp(v) ? 0 : 1; }\n')
    libdir = os.path.join(ROOT, "machineboss_amd")
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "cxx"), str(src), "-o", exe,
                           "-L", libdir, "-lmbhip", "-Wl,-rpath," + libdir])
    assert subprocess.run([exe]).returncode == 0


def test_count_programs_of_the_tiled_family_generate(tmp_path, monkeypatch):
    """Host only (mb_debug_jit_source needs no device): the two count programs of the tiled family for psw2dna.  FLAT (round 4): the
    closure Forward rounds carry no usage terms, ONE usage pass follows them whose slots cover the transitions that apply to a
    cell (input-token 1280 / 20, output-token 258 / 4, silent 146 over 32 lanes: 2 + 3 + 5), every transition of the machine sits
    in exactly one usage record.  LEVELLED (MB_MEDIUM_COUNT_FLAT=0): the exact program, usage terms from the log-sum-exp's own
    exponentials times one exp(max + B - LL) per state."""
    from machineboss_amd import capi
    m = Machine.fromFile(golden_path("preset", "psw2dna.json"))
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    flat = str(tmp_path / "flat.hip")
    capi.debug_jit_source(em, flat, mode=3 + 16, closure=2, G=2)
    src = open(flat).read()
    # round 5: the emitting transitions' usage is FUSED into the fill's emit rounds (same candidates, already in registers); the pass
    # that follows the fill keeps the silent transitions' 5 slots
    assert "#define JFLAT 1" in src and "usage pass: 5 slot(s)" in src and "sB_" not in src
    body = src[src.index("usage pass: 5 slot(s)"):]
    assert body.count("ex2(x") == 5                                    # one exponential per usage slot
    monkeypatch.setenv("MB_MEDIUM_COUNT_FUSE", "0")                    # round 4's form: every transition in the usage pass (2 + 3 + 5 slots)
    capi.debug_jit_source(em, flat, mode=3 + 16, closure=2, G=2)
    src = open(flat).read()
    assert "usage pass: 10 slot(s)" in src and src[src.index("usage pass: 10 slot(s)"):].count("ex2(x") == 10
    monkeypatch.delenv("MB_MEDIUM_COUNT_FUSE")
    monkeypatch.setenv("MB_MEDIUM_COUNT_FLAT", "0")
    lev = str(tmp_path / "lev.hip")
    capi.debug_jit_source(em, lev, mode=3 + 16, closure=2, G=2)
    src0 = open(lev).read()
    assert "#define JFLAT 0" in src0 and "usage pass" not in src0 and "sB_" in src0 and "ex2(mx_" in src0


def test_flat_usage_pass_goes_in_batches(tmp_path, monkeypatch):
    """A dense 100-state machine at 16 columns per wavefront (four lanes per column): 122 usage slots, emitted in batches of
    MB_JIT_FLAT_CHUNK (24) so that a batch's records and terms are what is live, not the whole pass (mb_medium_jit.cpp; as ONE batch
    the kernel spilled 600-800 VGPRs and its counts were wrong on the device, tests/test_gpu_parity.py).  The cross-compiled kernel
    of the batched pass keeps within the register file of a one-wavefront workgroup."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from randmachine import random_machine
    from machineboss_amd import capi
    em = random_machine(100, 1, 2, 46019, density=2.5, silent_density=1.5)
    monkeypatch.setenv("MB_JIT_REGBUDGET", "0")
    monkeypatch.setenv("MB_MEDIUM_COUNT_FUSE", "0")      # every transition in the usage pass (round 4's form: the longest pass; round 5 fuses the emitting ones into the fill)
    one = str(tmp_path / "batched.hip")
    capi.debug_jit_source(em, one, mode=3, closure=8, G=16)
    src = open(one).read()
    assert "usage pass: 122 slot(s)" in src
    body = src[src.index("usage pass: 122 slot(s)"):]
    assert body.count("ex2(x") == 122 and body.count("\n        {\n") == 6
    asm = str(tmp_path / "batched.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics", "-include", "hip/hip_runtime.h",
                           "-x", "hip", "--cuda-device-only", "-S", "-o", asm, one], stderr=subprocess.DEVNULL, timeout=600)
    spill = [ln for ln in open(asm) if ".vgpr_spill_count" in ln]
    assert spill and int(spill[0].split(":")[1]) == 0, spill
    monkeypatch.setenv("MB_JIT_FLAT_CHUNK", "0")
    capi.debug_jit_source(em, one, mode=3, closure=8, G=16)
    assert open(one).read().count("\n        {\n") == 0


def test_vector_issue_model_runs_on_a_generated_kernel(tmp_path):
    """scripts/valu_model.py end to end without a GPU: generate the rolling-Forward tile kernel of psw2dna,
    cross-compile it to gfx950 ISA, count the step loop -- a few hundred vector instructions for 4 x 271 cells, i.e. a quarter of
    an issue slot per cell (profiles/r04_valu_model.json: 0.26)."""
    import json
    import subprocess
    import sys
    from machineboss_amd import capi
    m = Machine.fromFile(golden_path("preset", "psw2dna.json"))
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    d = tmp_path / "jit"; d.mkdir()
    capi.debug_jit_source(em, str(d / "psw2dna.sum.tiles.fwd.clos.hip"), mode=16, closure=2, G=4)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "valu_model.py"), str(d), str(d / "model.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    k = json.load(open(d / "model.json"))["kernels"]["psw2dna.sum.tiles.fwd.clos.hip"]
    assert k["cells_per_loop"] == 4 * 271 and 150 <= k["loop"]["valu"] <= 400 and 0.15 <= k["issue_slots_per_cell"] <= 0.45


def test_compiled_weights_give_the_bits_of_the_scalar_evaluation():
    """EvaluatedMachine.reweighted (an EM iteration's re-evaluation: the machine's weight expressions as one flat program, compiled once,
    evaluated level by level with numpy) against EvaluatedMachine.fromMachine (evalWeight per transition, src/weight.cpp:241-300):
    the same bits on presets, on a composition (shared factors, function definitions), under several parameter sets; an undefined
    parameter and a division by zero raise as the scalar evaluation does; a transition given another weight object is seen."""
    import numpy as np
    from machineboss_amd import algebra as A
    from machineboss_amd.machine import Machine, MachineError
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(2)
    machines = [P("protpsw"), P("dnapsw"), P("psw2dna"), A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])]
    rng = np.random.RandomState(4)
    for m in machines:
        base = m.getParamDefs(True)
        e0 = EvaluatedMachine.fromMachine(m, base)
        for trial in range(3):
            p = {k: (float(rng.uniform(0.02, 0.98)) if isinstance(v, float) and 0.0 < v < 1.0 and trial else v) for k, v in base.items()}
            a = EvaluatedMachine.fromMachine(m, p).logWeight; b = e0.reweighted(m, p).logWeight
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    m = machines[0]
    e0 = EvaluatedMachine.fromMachine(m, m.getParamDefs(True))
    name = next(k for k, v in m.getParamDefs(True).items() if isinstance(v, float))
    broken = {k: v for k, v in m.getParamDefs(True).items()}
    t = m.state[0].trans[0]; keep = t.weight
    t.weight = {"/": [1.0, {"-": [name, name]}]}
    with pytest.raises(ZeroDivisionError): EvaluatedMachine.fromMachine(m, broken)
    with pytest.raises(ZeroDivisionError): e0.reweighted(m, broken)
    t.weight = "no_such_parameter"
    with pytest.raises(MachineError): e0.reweighted(m, broken)
    t.weight = keep
    assert np.array_equal(e0.reweighted(m, broken).logWeight.view(np.uint64), e0.logWeight.view(np.uint64))


def test_compiled_objective_matches_the_scalar_m_step(monkeypatch):
    """MachineObjective through the compiled program (evalmachine.CompiledWeights.objective: values forward, adjoints backward) against the
    scalar evaluation (evalWeight / symbolic derivatives per transition, src/counts.cpp:122-170): the objective to 1e-12, the gradient
    against central differences, and the same optimum from BFGS both ways on a composition whose weights are sums of products (no closed
    form)."""
    import numpy as np
    from machineboss_amd import algebra as A, fitter as F
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import CompiledWeights
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    m = A.composeLeftToRight([P("simple_introns"), P("translate"), P("dnapsw")])      # (354 states, 873 transitions, 84 parameters)

    class Counts: pass
    c = Counts(); c._flat = np.random.RandomState(3).uniform(0.0, 4.0, m.nTransitions()); c._flat[::7] = 0.0
    obj = F.MachineObjective(m, c, F.Constraints(), {})
    seed = obj.constraints.defaultParams()
    assert obj._closed_form(seed) is None
    cw = CompiledWeights(m, expand=obj.constantDefs, keep=obj.free)
    cvec = np.array([x for x, _ in obj.terms])
    rng = np.random.RandomState(5)
    for trial in range(3):
        p = {k: (float(rng.uniform(0.1, 0.9)) if trial else v) for k, v in seed.items()}
        defs = dict(obj.constantDefs); defs.update(p)
        E, dE = cw.objective(cvec, defs)
        ref = obj.value(p)
        assert abs(E - ref) <= 1e-12 * abs(ref)
        for name in list(dE)[::9]:
            d2 = dict(defs); hh = 1e-6 * max(abs(defs[name]), 1e-3)
            d2[name] = defs[name] + hh; Ep, _ = cw.objective(cvec, d2, False)
            d2[name] = defs[name] - hh; Em, _ = cw.objective(cvec, d2, False)
            fd = (Ep - Em) / (2 * hh)
            assert abs(dE[name] - fd) <= 1e-5 * max(abs(fd), 1.0), (name, dE[name], fd)
    a = obj.optimize(seed)
    monkeypatch.setenv("MB_FITTER_COMPILED", "0")
    b = obj.optimize(seed)
    assert abs(obj.value(a) - obj.value(b)) <= 1e-6 * abs(obj.value(b))
    for k in a:
        if isinstance(a[k], float): assert abs(a[k] - b[k]) <= 2e-4, (k, a[k], b[k])

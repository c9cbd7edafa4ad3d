"""GPU parity tests (-m gpu): the HIP engine, called through the C-ABI, against the CPU oracle on the same inputs.

Bars: Viterbi matrices, log-likelihoods and tracebacks are BIT-EXACT (integer/index work + one rounded fp64 add per
candidate); Forward/Backward cells and log-likelihoods agree with the oracle's exact-logsumexp mode to REL_EXACT and
with the reference's table mode to REL_TABLE (the north-star tolerance is 1e-4 relative; both bars are far tighter);
posterior counts agree to COUNT_TOL (summation order differs: fp64 atomics).
"""
import json
import math
import time
import os

import numpy as np
import pytest

from conftest import golden_path, load_json, load_matrix_json
from machineboss_amd.seqgen import synth_batch, synth_tokens

pytestmark = pytest.mark.gpu

REL_EXACT = 1e-11   # device exp/log1p vs libm, accumulated over the lattice
FAST_REL = 2e-6     # tiled families: fp32 exp/log correction term, ~1e-7 abs per cell, accumulated along the lattice
FAST_ABS = 2e-5


def _kn(name):
    """kernel name with the sweep GENERATED for the machine (k_wide_jit, mb_wide_jit.cpp) spelled like the interpreter it replaces"""
    return name.replace("k_wide_jit", "k_wide_retimed")


def _one_wg(name):
    """kernel name without the ' in k parts' of a sweep that ran k workgroups per sequence"""
    import re
    return re.sub(r" in \d+ parts", "", _kn(name))
ABS_TABLE = 1e-4    # vs the reference's table build: the table drops terms >= 10 nats below the running max and
REL_TABLE = 1e-4    # interpolates at step 1e-4 (src/logsumexp.h:20-21,48-70); 1e-4 relative is the north-star tolerance
COUNT_TOL = 1e-9
COUNT50_REL = 5e-6  # posterior counts of a 50 000-column one-tape sweep, PER TRANSITION against the exact oracle.  Round 5: the fills of an E-step over
                    # sequences of >= 10 000 symbols carry their log-sum-exp correction term in fp64 (mb_wide.hip wide_exp64): measured 4e-7 ... 8e-7 on three
                    # sequences / two parameter sets (scripts/count_accuracy_onetape.py).  With the fp32 term (MB_ONETAPE_COUNT_FP64=0, round 4) 6.4e-5 - 7.7e-5:
                    # a per-column error that REPEATS in stationary states and grows linearly, not a random walk; the reference's own default build
                    # (table-interpolated log-sum-exp) is 9.6e-2 from the exact oracle


@pytest.fixture(scope="module")
def capi():
    from machineboss_amd import capi as c
    if c.device_count() == 0:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return c


def close(a, b, rel, abs_=0.0):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    both_ninf = np.isneginf(a) & np.isneginf(b)
    fin = np.isfinite(a) & np.isfinite(b)
    if not np.all(both_ninf | fin):
        return False
    return bool(np.all(np.abs(a[fin] - b[fin]) <= abs_ + rel * np.abs(b[fin])))


CASES = [  # (preset?, name, params, inLen, outLen)
    (False, "bitnoise", "io", 3, 3),
    (False, "bitstutter-noise", "io", 5, 9),
    (True, "dnapsw", None, 37, 53),
    (True, "protpsw", None, 50, 50),
    (True, "psw2dna", None, 11, 40),
    (True, "translate", None, 6, 25),
]


def setup_case(machines, case, seed=7):
    preset, name, params, il, ol = case
    if params == "io":
        m, em = machines(name, load_json("io", "params.json"))
    else:
        m, em = machines(name, None, useDefaults=True, preset=preset)
    i, o = synth_tokens(seed, il, ol, em.nInTok, em.nOutTok)
    return m, em, i, o


@pytest.mark.parametrize("family", ["generic", "auto", "medium"])
@pytest.mark.parametrize("case", CASES, ids=[c[1] for c in CASES])
def test_fill_matrices(capi, oracle_mod, machines, case, family):
    """Full matrices vs the oracle.  The generic family evaluates log(1+exp(-x)) in fp64 (REL_EXACT); the tiled
    families evaluate that correction term in fp32 (FAST_REL / FAST_ABS, still ~3 orders inside the 1e-4 bar)."""
    m, em, i, o = setup_case(machines, case)
    om = oracle_mod.OracleMachine(em)
    # "auto": machines of <= 16 states take the small-machine family (mb_small.cpp); "medium" keeps the tiled family covered on them
    capi.set_kernel({"generic": capi.KERNEL_GENERIC, "auto": capi.KERNEL_AUTO, "medium": capi.KERNEL_MEDIUM}[family])
    try:
        dm = capi.DeviceMachine(em)
        assert list(dm.edge_order(0)) == list(om.incoming_order()) and list(dm.edge_order(1)) == list(om.outgoing_order())
        V = dm.fill(capi.MB_VITERBI, i, o)
        F = dm.fill(capi.MB_FORWARD, i, o)
        B = dm.fill(capi.MB_BACKWARD, i, o)
        fast = "generic" not in capi.last_kernel_name()
    finally:
        capi.set_kernel(capi.KERNEL_AUTO)
    rel, abs_ = (FAST_REL, FAST_ABS) if fast else (REL_EXACT, 0.0)
    assert np.array_equal(V, om.viterbi(i, o))        # Viterbi: bit-exact in every family
    assert close(F, om.forward(i, o, oracle_mod.SUM_EXACT), rel, abs_)
    assert close(F, om.forward(i, o, oracle_mod.SUM_TABLE), REL_TABLE, ABS_TABLE)
    assert close(B, om.backward(i, o, oracle_mod.SUM_EXACT), rel, abs_)
    assert close(B[0, 0, 0], F[-1, -1, -1], 1e-10, abs_)   # posterior LL = Forward LL (js/webgpu/test/test-cpu.mjs invariant)
    assert np.all(V <= F + 1e-6)                       # Viterbi <= Forward


def test_forward_start_state(capi, oracle_mod, machines):
    """ForwardMatrix 4-argument constructor: caller-chosen start state (src/forward.defs.h:16-21,36)."""
    m, em, i, o = setup_case(machines, CASES[2])
    F = capi.DeviceMachine(em).fill(capi.MB_FORWARD, i, o, startState=2)
    fast = "generic" not in capi.last_kernel_name()
    assert close(F, oracle_mod.OracleMachine(em).forward(i, o, oracle_mod.SUM_EXACT, startState=2),
                 FAST_REL if fast else REL_EXACT, FAST_ABS if fast else 0.0)


def test_reference_goldens_through_gpu(capi, machines):
    """The reference's own expected outputs, reproduced by the HIP path (Makefile:493-522,567-572)."""
    from machineboss_amd.dp import ForwardMatrix, BackwardMatrix, ViterbiMatrix, MachineCounts, SeqPair
    p = load_json("io", "params.json")
    m, em = machines("bitnoise", p)
    sp = SeqPair.fromJson(load_json("io", "tiny.json"))
    fwd, bwd = ForwardMatrix(em, sp), BackwardMatrix(em, sp)
    for mat, name in ((fwd, "fwd"), (bwd, "back")):
        for (ip, op, _), v in load_matrix_json("expect", name + "-bitnoise-params-tiny.json").items():
            got = mat.cell(ip, op, 0)
            assert (got == v) if not math.isfinite(v) else float("%.5g" % got) == v
    mc = MachineCounts(em, [sp])
    assert [[float("%.6g" % x) for x in row] for row in mc.count] == load_json("expect", "fwdback-bitnoise-params-tiny.json")
    mc2 = MachineCounts(em, [SeqPair(list("101"), list("001"))])
    assert {k: float("%.6g" % v) for k, v in mc2.paramCounts(m, p).items()} == load_json("expect", "counts.json")
    # Viterbi alignment through silent states
    m2, em2 = machines("bitstutter-noise", p)
    sp2 = SeqPair.fromJson(load_json("io", "difflen.json")[0])
    path = ViterbiMatrix(em2, sp2).path(m2)
    exp = load_json("expect", "align-stutter-noise-difflen.json")[0]["meta"]["path"]["trans"]
    got = [dict({"to": t.dest}, **({"in": t.inp} if t.inp else {}), **({"out": t.out} if t.out else {})) for t in path.trans]
    assert got == [{k: v for k, v in tr.items() if k in ("to", "in", "out")} for tr in exp]


@pytest.mark.parametrize("name,il,ol,n", [("dnapsw", 60, 45, 9), ("protpsw", 33, 41, 6), ("psw2dna", 9, 30, 5)])
def test_batch_api_ragged(capi, oracle_mod, machines, name, il, ol, n):
    """Batches with ragged lengths, including empty sequences on either tape."""
    m, em = machines(name, None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em)
    dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(3)
    pairs = []
    for k in range(n):
        a = 0 if k == 1 else int(rng.randint(1, il + 1)); b = 0 if k in (1, 2) else int(rng.randint(1, ol + 1))
        pairs.append(synth_tokens(100 + k, a, b, em.nInTok, em.nOutTok))
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    assert b.cells() == sum((len(x) + 1) * (len(y) + 1) * em.nStates for x, y in pairs)
    for flags in (capi.MB_MATERIALISE, capi.MB_ROLLING):
        ll = b.forward(flags)
        ref = [om.loglike(x, y, oracle_mod.SUM_EXACT) for x, y in pairs]
        fast = "generic" not in capi.last_kernel_name()
        assert close(ll, ref, FAST_REL if fast else 1e-10, FAST_ABS if fast else 0.0)
        assert close(ll, [om.loglike(x, y) for x, y in pairs], 1e-4)   # north-star tolerance vs the table build
    vll, off, edges = b.viterbi()
    for k, (x, y) in enumerate(pairs):
        V = om.viterbi(x, y)
        assert vll[k] == V[-1, -1, -1]                                   # bit-exact
        got = edges[off[k]:off[k + 1]]
        if V[-1, -1, -1] > -math.inf:
            assert np.array_equal(got, om.traceback(x, y, V))          # bit-exact traceback incl. tie-breaking
        else:
            assert len(got) == 0
    counts, s, cll = b.counts()
    ref_c = np.zeros(em.nTransitions); ref_s = 0.0
    for x, y in pairs:
        l = om.loglike(x, y, oracle_mod.SUM_EXACT)
        if l > -math.inf:
            ref_s += om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
        else:
            ref_s += l
    fast = "generic" not in capi.last_kernel_name()
    assert close(counts, ref_c, 1e-5 if fast else COUNT_TOL, 1e-7 if fast else 1e-12)
    assert (close(s, ref_s, FAST_REL if fast else 1e-10, FAST_ABS if fast else 0.0)) if math.isfinite(ref_s) else s == ref_s


def test_viterbi_ties_uniform_params(capi, oracle_mod, machines):
    """Uniform default parameters create many exact ties; tie-breaking must follow max_element (quirk Q4)."""
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em)
    dm = capi.DeviceMachine(em)
    x = np.ones(40, np.int32); y = np.ones(40, np.int32)   # homopolymers: every alignment of equal shape ties
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
    vll, off, edges = b.viterbi()
    V = om.viterbi(x, y)
    assert vll[0] == V[-1, -1, -1] and np.array_equal(edges, om.traceback(x, y, V))


def test_set_weights(capi, oracle_mod, machines):
    """mb_machine_set_weights: new parameters, same topology (one call per EM iteration, src/fitter.cpp:28-29)."""
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    x, y = synth_tokens(5, 30, 30, 4, 4)
    lw = em.logWeight.copy()
    lw[np.isfinite(lw)] *= 1.25
    dm.set_weights(lw)
    om = oracle_mod.OracleMachine(em.withLogWeights(lw))
    assert np.array_equal(dm.fill(capi.MB_VITERBI, x, y), om.viterbi(x, y))
    F = dm.fill(capi.MB_FORWARD, x, y)
    fast = "generic" not in capi.last_kernel_name()
    assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL if fast else REL_EXACT, FAST_ABS if fast else 0.0)


def test_errors(capi, machines):
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    with pytest.raises(capi.MbError, match="tokenize"):
        capi.DeviceBatch.from_pairs(dm, [(np.array([5], np.int32), np.array([1], np.int32))])
    with pytest.raises(capi.MbError, match="tokenize"):
        capi.DeviceBatch.from_pairs(dm, [(np.array([0], np.int32), np.array([1], np.int32))])
    # non-advancing machine is refused at creation (src/eval.cpp:44)
    bad = em.withLogWeights(em.logWeight)
    bad.src = em.src.copy(); bad.dst = em.dst.copy()
    e = int(np.where((em.inTok == 0) & (em.outTok == 0) & (em.src >= 1))[0][0])
    bad.dst[e] = bad.src[e]
    with pytest.raises(capi.MbError, match="topologically"):
        capi.DeviceMachine(bad)
    # empty batch is fine
    b = capi.DeviceBatch.from_pairs(dm, [])
    assert len(b.forward()) == 0


def test_unreachable_is_minus_infinity(capi, machines):
    """-L on a pair the machine cannot produce prints "-Infinity" (t/expect/tiny_uc_fail.json behaviour)."""
    m, em = machines("bitstutter-noise", load_json("io", "params.json"))
    dm = capi.DeviceMachine(em)
    x = em.inputTokenizer.tokenize(list("01")); y = np.zeros(0, np.int32)   # stutter machine must emit >= 1 per input
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
    assert b.forward()[0] == -math.inf
    vll, off, edges = b.viterbi()
    assert vll[0] == -math.inf and off[1] == 0


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_survey_anchors_gpu(capi, machines, idx):
    """Benchmark-scale reference outputs through the GPU path: Forward within 1e-4 relative
    (observed ~1e-8), Viterbi log-likelihood and path length exact at 10 significant digits."""
    a = load_json("survey_anchors.json")["anchors"][idx]
    m, em = machines(a["preset"], None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    i, o = synth_tokens(a["seed"], a["inLen"], a["outLen"], em.nInTok, em.nOutTok)
    b = capi.DeviceBatch.from_pairs(dm, [(i, o)])
    ll = b.forward()[0]
    assert abs(ll - a["forward"]) <= 1e-4 * abs(a["forward"])
    if a["forward_exact"] is not None:
        assert abs(ll - a["forward_exact"]) <= 2e-7 * abs(a["forward_exact"])   # fp32 correction terms accumulate along the lattice; device direct-logsumexp == the reference's -DLOG_SUM_EXP_SLOW build
    vll, off, edges = b.viterbi()
    assert float("%.10g" % vll[0]) == a["viterbi"] and off[1] == a["pathLen"]


# ---- tiled "lanes = states" family (mb_medium.hip) ---------------------------------------------------------------
def _medium_case(capi, oracle_mod, em, x, y, G, monkeypatch, jit=0):
    """jit=0: the ahead-of-time interpreter kernel k_medium_tile; jit=1: the hiprtc-specialised k_medium_jit."""
    monkeypatch.setenv("MB_MEDIUM_G", str(G))
    monkeypatch.setenv("MB_MEDIUM_JIT", str(jit))
    om = oracle_mod.OracleMachine(em)
    capi.set_kernel(capi.KERNEL_MEDIUM)
    try:
        dm = capi.DeviceMachine(em)
        V = dm.fill(capi.MB_VITERBI, x, y)
        F = dm.fill(capi.MB_FORWARD, x, y)
        B = dm.fill(capi.MB_BACKWARD, x, y)
        assert ("k_medium_jit" if jit else "k_medium_tile") in capi.last_kernel_name()
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        llr = b.forward(capi.MB_ROLLING)[0]
    finally:
        capi.set_kernel(capi.KERNEL_AUTO)
    assert np.array_equal(V, om.viterbi(x, y))                                   # bit-exact
    Fe = om.forward(x, y, oracle_mod.SUM_EXACT)
    assert close(F, Fe, FAST_REL, FAST_ABS)
    assert close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
    assert close(llr, Fe[-1, -1, -1], FAST_REL, FAST_ABS)
    assert close(llr, om.loglike(x, y), 1e-4)                                     # north-star bar vs the table build


@pytest.mark.parametrize("G", [1, 2, 4, 8])
@pytest.mark.parametrize("shape", [(5, 40), (70, 300), (0, 33), (41, 0), (130, 17)])
def test_medium_psw2dna(capi, oracle_mod, machines, monkeypatch, G, shape):
    """psw2dna (271 states, 10 silent levels): multi-strip (inLen > columns per workgroup) and multi-tile sweeps."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    x, y = synth_tokens(11, shape[0], shape[1], em.nInTok, em.nOutTok)
    _medium_case(capi, oracle_mod, em, x, y, G, monkeypatch)


@pytest.mark.parametrize("G,shape", [(2, (70, 300)), (2, (0, 33)), (4, (41, 0)), (1, (130, 17))])
def test_medium_psw2dna_jit(capi, oracle_mod, machines, monkeypatch, G, shape):
    """The same checks through the run-time specialised kernel (hiprtc)."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    x, y = synth_tokens(11, shape[0], shape[1], em.nInTok, em.nOutTok)
    _medium_case(capi, oracle_mod, em, x, y, G, monkeypatch, jit=1)


@pytest.mark.parametrize("G", [1, 4])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_medium_random_machines(capi, oracle_mod, monkeypatch, G, seed):
    """Random machines with match edges (3-slot ring), duplicate edges, several edges per label, -inf weights."""
    from randmachine import random_machine, random_seq
    S = [37, 90, 150][seed - 1]
    em = random_machine(S, 3, 4, seed, allow_inf=(seed == 2))
    rng = np.random.RandomState(seed)
    for il, ol in [(25, 31), (0, 9), (60, 45)]:
        _medium_case(capi, oracle_mod, em, random_seq(rng, il, 3), random_seq(rng, ol, 4), G, monkeypatch, jit=(seed + G) % 2)


def test_medium_batch_counts_and_paths(capi, oracle_mod, machines):
    """Viterbi tracebacks and posterior counts on top of medium-family matrices (ragged batch)."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(40 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate([(12, 50), (40, 90), (3, 7)])]
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    vll, off, edges = b.viterbi()
    assert "k_medium" in capi.last_kernel_name()
    for k, (x, y) in enumerate(pairs):
        V = om.viterbi(x, y)
        assert vll[k] == V[-1, -1, -1] and np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, V))
    counts, s, ll = b.counts()
    ref_c = np.zeros(em.nTransitions); ref_s = 0.0
    for x, y in pairs:
        ref_s += om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
    assert close(counts, ref_c, 1e-5, 1e-7) and close(s, ref_s, FAST_REL, FAST_ABS)


@pytest.mark.parametrize("case", ["psw2dna", "random-match-60", "random-300", "c4b"])
def test_three_pass_counts_match_the_fused_sweep_and_the_oracle(capi, oracle_mod, machines, monkeypatch, case):
    """BackwardMatrix::getCounts as a THIRD pass over two materialised matrices (mb_usage.hip, round 6; src/backward.cpp:58-87): one
    workgroup per input column, a lane per transition, both matrices streamed once.  Forced on (MB_MEDIUM_COUNT_PASSES=3) and off (2: the
    fused sweep) on psw2dna, on random machines with match transitions (both tapes consumed: B(i + 1, o + 1)) and of 300 states, and on
    the 482-state composition it was built for: the two agree, both agree with the oracle, ragged batches with empty sequences,
    MB_DETERMINISTIC=1 included; a weight update is followed."""
    from randmachine import random_machine, random_seq
    rng = np.random.RandomState(7)
    if case == "psw2dna":
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
        pairs = [synth_tokens(60 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate([(12, 50), (40, 90), (3, 7), (0, 9), (5, 0), (33, 130)])]
    elif case == "c4b":
        from machineboss_amd import algebra
        from machineboss_amd.evalmachine import EvaluatedMachine
        em = EvaluatedMachine.fromMachine(algebra.config4bMachine(golden_path("preset")), None, useDefaults=True)
        pairs = [synth_tokens(70 + k, a, b, em.nInTok, 3) for k, (a, b) in enumerate([(9, 40), (20, 70), (2, 5)])]
    else:
        S = 60 if case == "random-match-60" else 300
        em = random_machine(S, 3, 2, 4242 + S, density=2.0, silent_density=0.8, allow_inf=True)
        assert np.any((np.asarray(em.inTok) != 0) & (np.asarray(em.outTok) != 0))      # match transitions
        pairs = [(random_seq(rng, a, 3), random_seq(rng, b, 2)) for a, b in [(7, 11), (30, 44), (0, 6), (9, 0), (1, 1)]]
    om = oracle_mod.OracleMachine(em)
    ref_c = np.zeros(em.nTransitions); ref_s = 0.0
    for x, y in pairs: ref_s += om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
    got = {}
    for passes in ("3", "2"):
        monkeypatch.setenv("MB_MEDIUM_COUNT_PASSES", passes)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        counts, s, ll = b.counts()
        kern = capi.last_kernel_name()
        assert ("k_medium_usage" in kern) == (passes == "3"), kern
        assert close(counts, ref_c, 1e-5, 1e-7) and close(s, ref_s, FAST_REL, FAST_ABS)
        got[passes] = counts
        if passes == "3":
            monkeypatch.setenv("MB_DETERMINISTIC", "1")
            c1, _, _ = b.counts(); c2, _, _ = b.counts()
            assert np.array_equal(c1, c2) and close(c1, counts, 1e-8, 1e-9)
            monkeypatch.delenv("MB_DETERMINISTIC")
            # new weights: the usage records follow (half of every weight's probability)
            lw = np.asarray(em.logWeight) + np.log(0.5)
            dm.set_weights(lw)
            c3, s3, _ = b.counts()
            assert "k_medium_usage" in capi.last_kernel_name()
            import copy
            em2 = copy.copy(em); em2.logWeight = lw
            dm2 = capi.DeviceMachine(em2)
            c4, s4, _ = capi.DeviceBatch.from_pairs(dm2, pairs).counts()
            assert close(c3, c4, 1e-9, 1e-12) and close(s3, s4, 1e-12)
            dm2.close()
            dm.set_weights(np.asarray(em.logWeight))
        dm.close()
    assert close(got["3"], got["2"], 2e-6, 1e-8)
    # the default is the fused sweep (two matrices per pair halve the pairs of a chunk, and the fills of small chunks lose more than the usage pass gains)
    monkeypatch.delenv("MB_MEDIUM_COUNT_PASSES")
    if case in ("psw2dna", "c4b"):
        dm = capi.DeviceMachine(em)
        capi.DeviceBatch.from_pairs(dm, pairs).counts()
        assert "k_medium_usage" not in capi.last_kernel_name()


def test_tiled_family_placement_serves_every_strip_width(capi):
    """A 17-state machine whose rolling kernel (narrow strips of a short batch, tiles without a matrix) spills at the first
    register budget: the re-plan that follows is shared with the matrix kernel of the wider strips, which must still fit the
    LDS -- it silently ran the ahead-of-time interpreter (scripts/fuzz_gpu.py seed 80033, round 3).  Both calls run the
    specialised kernel and give the same bits."""
    from randmachine import random_machine, random_seq
    rng = np.random.RandomState(80033)
    S = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 16, 17, 33, 64, 100, 257, 300, 700])); nIn = int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
    em = random_machine(S, nIn, nOut, 80033, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=False)
    assert em.nStates == 17
    n = int(rng.randint(1, 6))
    pairs = [(random_seq(rng, int(rng.randint(0, 40)), nIn), random_seq(rng, int(rng.randint(0, 60)), nOut)) for _ in range(n)]
    dm = capi.DeviceMachine(em); b = capi.DeviceBatch.from_pairs(dm, pairs)
    ll = b.forward(capi.MB_ROLLING); k1 = capi.last_kernel_name()
    llm = b.forward(capi.MB_MATERIALISE); k2 = capi.last_kernel_name()
    assert k1 == k2 == "k_medium_jit" and np.array_equal(ll, llm)


@pytest.mark.parametrize("case", ["psw2dna", "random40", "random150", "random257", "random300split"])
def test_tiled_family_keeps_no_fp64_matrix(capi, oracle_mod, machines, case):
    """Round 3: on the tiled family, Viterbi keeps ONE traceback byte per cell (MED_MODE_TB: the winning candidate's table and
    index, walked by k_traceback_bytes) and the count sweep keeps NO Forward matrix (tiles that hand their ring state over
    through boundary records and halo columns of the few states other strips read).  Both against the oracle -- paths
    bit-exact, first-maximum tie-break included -- and against the round-2 paths (fp64 Viterbi matrix, Forward matrix stored),
    on lattices of several strips and several blocks, ragged, with empty sequences; H (states per halo row) from 3 to > 256;
    one machine whose exact program splits high-degree states (the byte sweep then runs an unsplit twin)."""
    from randmachine import random_machine, random_seq
    if case == "psw2dna":
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
        shapes = [(30, 90), (100, 300), (0, 5), (7, 0), (33, 257), (64, 129), (70, 700)]
        pairs = [synth_tokens(k + 1, il, ol, em.nInTok, em.nOutTok) for k, (il, ol) in enumerate(shapes)]
    else:
        S = int(case[6:9].rstrip("s"))
        rng = np.random.RandomState(S)
        em = random_machine(S, 3, 2, 100 + S, density=4.0 if "split" in case else 2.0, silent_density=2.5 if "split" in case else 1.0, dup=True)
        pairs = [(random_seq(rng, int(rng.randint(0, 90)), 3), random_seq(rng, int(rng.randint(0, 200)), 2)) for _ in range(5)] + [(random_seq(rng, 70, 3), random_seq(rng, 330, 2))]
    om = oracle_mod.OracleMachine(em)
    res = {}
    for new in ("1", "0"):
        capi.set_option("MB_MEDIUM_TB", new); capi.set_option("MB_MEDIUM_COUNTS_ROLL", new)
        try:
            dm = capi.DeviceMachine(em)
            b = capi.DeviceBatch.from_pairs(dm, pairs)
            v = b.viterbi(); kv = capi.last_kernel_name()
            c = b.counts(); kc = capi.last_kernel_name()
            res[new] = (v, c)
            assert kv == "k_medium_jit" and kc == "k_medium_jit"
            dm.close()
        finally:
            capi.set_option("MB_MEDIUM_TB", None); capi.set_option("MB_MEDIUM_COUNTS_ROLL", None)
    (v1, c1), (v0, c0) = res["1"], res["0"]
    assert np.array_equal(v1[0], v0[0]) and np.array_equal(v1[1], v0[1]) and np.array_equal(v1[2], v0[2])
    assert close(c1[0], c0[0], 1e-6, 1e-9) and close(c1[2], c0[2], 1e-9, 1e-9)
    ref_c = np.zeros(em.nTransitions)
    for k, (x, y) in enumerate(pairs):
        V = om.viterbi(x, y)
        assert v1[0][k] == V[-1, -1, -1]
        got = v1[2][v1[1][k]:v1[1][k + 1]]
        if V[-1, -1, -1] > -math.inf:
            assert np.array_equal(got, om.traceback(x, y, V))
        else:
            assert len(got) == 0
        if om.loglike(x, y, oracle_mod.SUM_EXACT) > -math.inf:
            om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
    assert close(c1[0], ref_c, 1e-5, 1e-7)


@pytest.mark.parametrize("flat", ["1", "0"])
def test_tiled_family_128_step_tiles_and_both_count_programs(capi, oracle_mod, machines, flat):
    """What sweeps of >= 4096 output positions select by themselves -- tiles of 128 steps (tile_steps, mb_medium.hip) -- forced onto
    lattices the oracle fills in a second (the 128-step tiles were never oracle-checked in traceback-byte or count
    mode), with the count program in both of its forms: FLAT (round 4: closure Forward rounds + one usage pass of one
    transition per lane) and LEVELLED (the exact program, usage terms from the log-sum-exp's own exponentials)."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    shapes = [(70, 700), (33, 257), (100, 300), (64, 129), (0, 5), (7, 0), (40, 1100)]
    pairs = [synth_tokens(k + 11, il, ol, em.nInTok, em.nOutTok) for k, (il, ol) in enumerate(shapes)]
    om = oracle_mod.OracleMachine(em)
    capi.set_option("MB_MEDIUM_TS", "128"); capi.set_option("MB_MEDIUM_COUNT_FLAT", flat)
    try:
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        vll, off, edges = b.viterbi(); assert capi.last_kernel_name() == "k_medium_jit"
        counts, s, cll = b.counts(); assert capi.last_kernel_name() == "k_medium_jit"
        llm = b.forward(capi.MB_MATERIALISE)
        dm.close()
    finally:
        capi.set_option("MB_MEDIUM_TS", None); capi.set_option("MB_MEDIUM_COUNT_FLAT", None)
    ref_c = np.zeros(em.nTransitions)
    for k, (x, y) in enumerate(pairs):
        V = om.viterbi(x, y)
        assert vll[k] == V[-1, -1, -1] and np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, V))
        ll = om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
        assert close([cll[k], llm[k]], [ll, ll], FAST_REL, FAST_ABS)
    assert close(counts, ref_c, 1e-5, 1e-7)


def test_flat_count_program_of_a_hundred_slots(capi, oracle_mod):
    """16 columns per wavefront on a dense 100-state machine: four lanes per column, so the usage pass of the flat count program
    has 126 slots.  Emitted as one batch it kept every record and term live at once (600-800 spilled VGPRs) and the sweep lost the
    Backward values of one group of four states (knob fuzz, seed 46000 case 19: counts of 27 transitions wrong by up to 7.7); the
    pass now goes in batches of MB_JIT_FLAT_CHUNK slots (mb_medium_jit.cpp).  Oracle: src/backward.cpp:58-87."""
    from randmachine import random_machine, random_seq
    em = random_machine(100, 1, 2, 46019, density=2.5, silent_density=1.5)
    rng = np.random.RandomState(7)
    pairs = [(random_seq(rng, il, 1), random_seq(rng, ol, 2)) for il, ol in ((31, 36), (29, 50), (0, 9), (17, 0), (40, 140), (33, 31), (8, 61))]
    om = oracle_mod.OracleMachine(em)
    pairs = [(x, y) for x, y in pairs if om.loglike(x, y, oracle_mod.SUM_EXACT) > -math.inf]     # (the oracle's counts of an impossible pair are NaN)
    assert len(pairs) >= 3
    ref = np.zeros(em.nTransitions); lls = [om.counts_add(x, y, ref, oracle_mod.SUM_EXACT) for x, y in pairs]
    got = {}
    for G in ("16", "8"):
        capi.set_option("MB_MEDIUM_G", G)
        try:
            dm = capi.DeviceMachine(em)
            cnt, s, cll = capi.DeviceBatch.from_pairs(dm, pairs).counts(); assert capi.last_kernel_name() == "k_medium_jit"
            dm.close()
        finally:
            capi.set_option("MB_MEDIUM_G", None)
        assert close(cll, lls, FAST_REL, FAST_ABS)
        assert close(cnt, ref, 1e-5, 1e-7), (G, float(np.abs(cnt - ref).max()))


def _ram_gb():
    import psutil
    return psutil.virtual_memory().available / 1e9


@pytest.mark.parametrize("which", ["psw2dna", "c4b"])
def test_baseline_config4_one_pair_at_its_stated_size_against_the_oracle(capi, oracle_mod, machines, which):
    """BASELINE config 4 at its own shape -- one pair of a 487-residue protein x 10 kb of DNA -- against the ORACLE, not through
    properties (the oracle fills this lattice in a minute): Forward log-likelihood (rolling, materialised,
    the count sweep's own), the Viterbi score and the whole path bit for bit (one traceback byte per cell on the device, a
    10.6 GB fp64 matrix in the oracle), and the posterior count of every transition (src/forward.defs.h:23-49,
    src/viterbi.cpp:18-51, src/dpmatrix.defs.h:82-110, src/backward.cpp:58-87).  Both readings of the config: psw2dna (271
    states, the bench line) and the literal protpsw . translate . dnapsw (482 states, 22 silent levels; counts at 2 kb -- its
    two oracle matrices at 10 kb are 38 GB)."""
    if which == "psw2dna":
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
        nOut = em.nOutTok
    else:
        from machineboss_amd import algebra
        from machineboss_amd.evalmachine import EvaluatedMachine
        em = EvaluatedMachine.fromMachine(algebra.config4bMachine(golden_path("preset")), None, useDefaults=True)
        assert em.nStates == 482 and em.nTransitions == 3095
        nOut = 3                                                   # DNA over {A,C,G}: no stop codons
    one = 488 * 10001 * em.nStates * 8 / 1e9
    inTok, inOff, outTok, outOff = synth_batch(4, 2, 487, 10000, em.nInTok, nOut)
    x, y = inTok[inOff[1]:inOff[2]], outTok[outOff[1]:outOff[2]]
    yc = y if which == "psw2dna" else y[:2000]
    two = 2 * 488 * (len(yc) + 1) * em.nStates * 8 / 1e9
    if _ram_gb() < one + two + 8: pytest.skip("host memory: the oracle's Viterbi matrix of this lattice is %.1f GB, its Forward and Backward matrices %.1f GB" % (one, two))
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    om.loglike(x[:3], y[:3], oracle_mod.SUM_TABLE)      # (the table is built by the first call that wants it)
    # the oracle's four sweeps are independent and single-threaded (ctypes releases the GIL): side by side, two minutes become one
    from concurrent.futures import ThreadPoolExecutor
    ref_c = np.zeros(em.nTransitions)
    def vit():
        V = om.viterbi(x, y)
        return V[-1, -1, -1], om.traceback(x, y, V)
    with ThreadPoolExecutor(4) as pool:
        f_ref = pool.submit(om.loglike, x, y, oracle_mod.SUM_EXACT); f_tab = pool.submit(om.loglike, x, y, oracle_mod.SUM_TABLE)
        f_vit = pool.submit(vit); f_cnt = pool.submit(om.counts_add, x, yc, ref_c, oracle_mod.SUM_EXACT)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        llr = b.forward(capi.MB_ROLLING); llm = b.forward(capi.MB_MATERIALISE)
        assert capi.last_kernel_name() == "k_medium_jit"
        vll, off, edges = b.viterbi(); assert capi.last_kernel_name() == "k_medium_jit"
        bc = capi.DeviceBatch.from_pairs(dm, [(x, yc)])
        counts, s, cll = bc.counts(); assert capi.last_kernel_name() == "k_medium_jit"
        ref = f_ref.result()
        assert close([llr[0], llm[0]], [ref, ref], FAST_REL, FAST_ABS)
        assert close(llr, [f_tab.result()], REL_TABLE, ABS_TABLE)      # the reference's default build, at the north-star tolerance
        vEnd, want = f_vit.result()
        assert vll[0] == vEnd
        assert np.array_equal(edges[off[0]:off[1]], want)
        # counts: every transition
        llc = f_cnt.result()
    assert close(cll, [llc], FAST_REL, FAST_ABS)
    assert close(counts, ref_c, 1e-5, 1e-7)
    dm.close()


def test_baseline_config5_one_sequence_at_50kb_against_the_oracle(capi, oracle_mod):
    """BASELINE config 5 at its stated length against the ORACLE (a 50 kb sequence costs the oracle seconds on
    the 5 063-state machine, not minutes): log-likelihood through every route -- cut in two and joined, the plain retimed sweep,
    the materialised fill -- against the exact and the table build; the Viterbi MATRIX bit for bit (2 GB) and the path; the
    posterior count of every transition (src/forward.defs.h:23-49, src/viterbi.cpp:18-51, src/dpmatrix.defs.h:82-110,
    src/backward.cpp:58-87, src/counts.cpp:57-64)."""
    L = 50000
    m, em = _profile_machine(20)
    assert em.nInTok == 0 and em.nStates == 5063
    if _ram_gb() < 14: pytest.skip("host memory: two oracle matrices of 2 GB each, the device's copies and the comparison")
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    x = np.zeros(0, np.int32)
    y = np.random.RandomState(2050).randint(1, 4, size=L).astype(np.int32)      # DNA over {A,C,G}: no stop codons
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
    ref = om.loglike(x, y, oracle_mod.SUM_EXACT)
    llr = b.forward(capi.MB_ROLLING); k1 = capi.last_kernel_name()
    capi.set_option("MB_ONETAPE_SPLIT", "0")
    try:
        llp = b.forward(capi.MB_ROLLING); k2 = capi.last_kernel_name()
    finally:
        capi.set_option("MB_ONETAPE_SPLIT", None)
    llm = b.forward(capi.MB_MATERIALISE)
    assert _one_wg(k1) == "k_wide_retimed<0> x2 + k_onetape_join" and _one_wg(k2) == "k_wide_retimed<0>"
    assert close([llr[0], llp[0], llm[0]], [ref] * 3, FAST_REL, FAST_ABS)
    assert close(llr, [om.loglike(x, y, oracle_mod.SUM_TABLE)], REL_TABLE, ABS_TABLE)
    V = om.viterbi(x, y)
    Vd = dm.fill(capi.MB_VITERBI, x, y); assert _kn(capi.last_kernel_name()).startswith("k_wide_retimed<1")
    assert np.array_equal(Vd, V)
    Vd_end = float(Vd[-1, -1, -1])
    del Vd
    want = om.traceback(x, y, V)
    del V
    for tb in ("0", "1"):      # the path from the fp64 matrix, and from one traceback code per cell (k_wide_retimed<1,codes> + k_onetape_traceback_codes)
        capi.set_option("MB_ONETAPE_TB", tb)
        try:
            vll, off, edges = b.viterbi()
        finally:
            capi.set_option("MB_ONETAPE_TB", None)
        assert ("codes" in capi.last_kernel_name()) == (tb == "1")
        assert vll[0] == Vd_end and np.array_equal(edges[off[0]:off[1]], want)
    counts, s, cll = b.counts()
    ref_c = np.zeros(em.nTransitions)
    llc = om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
    assert close(cll, [llc], FAST_REL, FAST_ABS)
    # per transition: the fp32 correction terms of 50 000 columns of log-sum-exp (1e-7 each) walk at random in F and B
    dev = np.abs(counts - ref_c) / np.maximum(np.abs(ref_c), 1e-3)
    print("config 5 at 50 kb: largest per-transition count deviation %.3g (relative, counts below 1e-3 taken as 1e-3)" % dev.max())
    assert close(counts, ref_c, COUNT50_REL, 1e-3 * COUNT50_REL)
    dm.close()


@pytest.mark.parametrize("L,seed,params", [(50000, 7, "random"), (20000, 11, "uniform"), (12000, 3, "random"), (9000, 5, "uniform")])
def test_one_tape_counts_of_long_sequences_against_the_oracle(capi, oracle_mod, L, seed, params):
    """the per-transition comparison of the one-tape E-step with the EXACT oracle on more than one sequence and
    on a NON-UNIFORM parameter set (every norm group a random point of its simplex, every prob in (0.05, 0.95): no ties, another
    dynamic range), at 50 kb and at shorter lengths either side of the 10 000-symbol threshold from which the fills carry their
    correction term in fp64 (src/backward.cpp:58-87, src/logsumexp.h:72-90).  The bound is 5e-6 (1e-4 asked; measured 4e-7 ... 8e-7 with
    the fp64 term, <= 4.9e-5 below the threshold with the fp32 term at 9 000 symbols -- bound 1e-4 there)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import count_accuracy_onetape as cao
    from machineboss_amd.evalmachine import EvaluatedMachine
    if _ram_gb() < 14 and L > 20000: pytest.skip("host memory: two oracle matrices of 2 GB each")
    m = cao.profile_machine(20)
    em = EvaluatedMachine.fromMachine(m, cao.random_params(m, 99)) if params == "random" else EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    x = np.zeros(0, np.int32); y = np.random.RandomState(seed).randint(1, 4, size=L).astype(np.int32)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
    counts, s, ll = b.counts()
    ref = np.zeros(em.nTransitions); llo = om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
    dev = np.abs(counts - ref) / np.maximum(np.abs(ref), 1e-3)
    print("one-tape E-step, L = %d, %s parameters: largest per-transition count deviation %.3g, log-likelihood %.3g relative" % (L, params, dev.max(), abs(ll[0] - llo) / abs(llo)))
    bound = COUNT50_REL if L >= 10000 else 1e-4
    assert close(counts, ref, bound, 1e-3 * bound) and abs(ll[0] - llo) <= (1e-12 if L >= 10000 else 1e-7) * abs(llo)
    assert abs(counts[np.asarray(em.outTok) != 0].sum() / L - 1.0) < 1e-6
    b.close(); dm.close()


@pytest.mark.parametrize("name", ["psw2dna", "random150", "random257"])
def test_in_place_ring_of_the_matrix_free_sum_kernels(capi, oracle_mod, machines, monkeypatch, name):
    """MB_MEDIUM_INPLACE_RING=1 (built and measured in round 5; off by default because it does not pay): the
    matrix-free sum kernels keep ONE full vector per column -- read as the step before by the emit rounds of stage 0, whose loads all
    precede the stage's first store, and overwritten in place -- plus short vectors of the halo states, and run as many wavefronts
    as that leaves room for.  Log-likelihoods through the tile pipeline (few pairs, blocks of 64 steps: boundary records) and
    through the strip sweep, full and restricted envelopes, against the oracle and against the plain ring, bit for bit."""
    from randmachine import random_machine
    from machineboss_amd.seqpair import Envelope, SeqPair
    if name == "psw2dna":
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    else:
        em = random_machine(int(name[6:]), 3, 3, 77 + int(name[6:]), density=1.6, silent_density=1.2)
    om = oracle_mod.OracleMachine(em)
    rng = np.random.RandomState(5)
    shapes = [(70, 150), (33, 95), (0, 40), (64, 1), (90, 130)] if name == "psw2dna" else [(40, 90), (17, 70), (0, 30), (50, 2)]
    pairs = [(rng.randint(1, em.nInTok + 1, size=il).astype(np.int32), rng.randint(1, em.nOutTok + 1, size=ol).astype(np.int32)) for il, ol in shapes]
    ref = np.array([om.loglike(x, y, oracle_mod.SUM_EXACT) for x, y in pairs])
    got = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("MB_MEDIUM_INPLACE_RING", ring)
        monkeypatch.setenv("MB_MEDIUM_TS", "64")
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        monkeypatch.setenv("MB_ROLLING_MIN_PAIRS", "100000")      # few pairs: tiles without a matrix (boundary records between blocks of 64 steps)
        tiles = b.forward(capi.MB_ROLLING); k1 = capi.last_kernel_name()
        monkeypatch.setenv("MB_ROLLING_MIN_PAIRS", "0")           # the strip sweep
        strips = b.forward(capi.MB_ROLLING)
        assert k1 == "k_medium_jit"
        got[ring] = (tiles, strips)
        if ring == "1": n_inplace = int(capi.load().mb_get_option(b"MB_INFO_INPLACE_RING_KERNELS") or 0)
        for v in (tiles, strips):
            assert close(v, ref, FAST_REL, FAST_ABS)
        b.close(); dm.close()
    assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])      # same rounds, same arithmetic: same bits
    assert n_inplace >= (2 if name == "psw2dna" else 0)      # (psw2dna: both kinds took it -- 3 halo states of 271; a dense random machine may read most states across columns)


def test_tiled_family_byte_sweep_under_envelopes(capi, oracle_mod, machines):
    """The same two sweeps with restricted envelopes: tiles without a cell of the envelope do not run (their halo rows read
    -inf, the next block of the strip starts from -inf instead of a boundary record), among them a gapless stretch that
    crosses strip and block boundaries on its diagonal (the case of test_gapless_path_envelope_... for this family)."""
    from machineboss_amd.seqpair import Envelope
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(5)
    for il, ol, width in [(40, 130, 0), (70, 260, 3), (100, 100, 0)]:
        x, y = synth_tokens(900 + il, il, ol, em.nInTok, em.nOutTok)
        # an alignment that consumes three output symbols per input symbol where it can (protein -> DNA), gaps elsewhere
        cols = []; i = o = 0
        while i < il or o < ol:
            if i < il and o + 3 <= ol and rng.rand() < 0.8: cols += [("a", "")] + [("", "b")] * 3; i += 1; o += 3
            elif o < ol: cols.append(("", "b")); o += 1
            else: cols.append(("a", "")); i += 1
        env = Envelope.pathAreaEnvelope(cols, width)
        with oracle_mod.envelope(env.inStart, env.inEnd):
            Vo = om.viterbi(x, y)
            po = om.traceback(x, y, Vo) if Vo[-1, -1, -1] > -math.inf else None
            ref_c = np.zeros(em.nTransitions)
            ref_ll = om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT) if om.loglike(x, y, oracle_mod.SUM_EXACT) > -math.inf else -math.inf
        b = capi.DeviceBatch.from_pairs(dm, [(x, y), (x[:20], y[:50])])
        b.set_envelopes([(env.inStart, env.inEnd), None])
        vll, off, edges = b.viterbi()
        assert capi.last_kernel_name() == "k_medium_jit" and vll[0] == Vo[-1, -1, -1]
        if po is not None:
            assert np.array_equal(edges[off[0]:off[1]], po)
        b1 = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        b1.set_envelopes([(env.inStart, env.inEnd)])
        counts, s_ll, cll = b1.counts()
        if math.isfinite(ref_ll):
            assert close(counts, ref_c, 1e-5, 1e-7) and close(cll[0], ref_ll, FAST_REL, FAST_ABS)
        else:
            assert cll[0] == -math.inf and not counts.any()


def test_medium_vs_generic_large(capi, machines):
    """Size-independent check at a larger shape: both kernel families fill identical Viterbi matrices and agree on
    Forward; rolling == materialised log-likelihood."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    x, y = synth_tokens(77, 200, 1500, em.nInTok, em.nOutTok)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)] * 3)
    llm = b.forward(capi.MB_MATERIALISE); llr = b.forward(capi.MB_ROLLING)
    vm, _, _ = b.viterbi(paths=False)
    capi.set_kernel(capi.KERNEL_GENERIC)
    try:
        llg = b.forward(capi.MB_MATERIALISE); vg, _, _ = b.viterbi(paths=False)
    finally:
        capi.set_kernel(capi.KERNEL_AUTO)
    assert np.array_equal(vm, vg) and np.all(vm == vm[0])
    assert close(llm, llg, FAST_REL, FAST_ABS) and close(llr, llg, FAST_REL, FAST_ABS) and np.all(llm == llm[0])


def test_cxx_facade(capi, tmp_path):
    """The C++ shim with the reference's class names (machineboss_amd/cxx/mb_dp.hpp) over the C-ABI, on the
    reference's bitnoise golden case (t/expect/{fwd,back,fwdback}-bitnoise-params-tiny.json)."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "test_facade")
    libdir = os.path.join(ROOT, "machineboss_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "cxx"),
                           os.path.join(ROOT, "tests", "cxx", "test_facade.cpp"), "-o", exe, "-L", libdir, "-lmbhip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "FACADE OK" in out.stdout, out.stdout + out.stderr


# ---- host-side path functions over device-filled matrices ------
def _check_path(em, m, path, x, y, startState=0, endState=None):
    """A MachinePath must be contiguous, start/end at the right states and spell both sequences."""
    s = startState; ins = []; outs = []
    for (src, ti), tr in zip(path.steps, path.trans):
        assert src == s and m.state[src].getTransition(ti) is tr
        if tr.inp: ins.append(tr.inp)
        if tr.out: outs.append(tr.out)
        s = tr.dest
    assert s == (em.nStates - 1 if endState is None else endState)
    assert list(em.inputTokenizer.tokenize(ins)) == list(x) and list(em.outputTokenizer.tokenize(outs)) == list(y)


def test_write_json_matches_reference_text(capi, machines):
    """DPMatrix::writeJson (src/dpmatrix.defs.h:39-53) reproduces the reference's golden matrix dumps byte for byte."""
    from machineboss_amd.dp import ForwardMatrix, BackwardMatrix, SeqPair
    m, em = machines("bitnoise", load_json("io", "params.json"))
    sp = SeqPair.fromJson(load_json("io", "tiny.json"))
    assert ForwardMatrix(em, sp).writeJson() == open(golden_path("expect", "fwd-bitnoise-params-tiny.json")).read()
    assert BackwardMatrix(em, sp).writeJson() == open(golden_path("expect", "back-bitnoise-params-tiny.json")).read()


@pytest.mark.parametrize("name,il,ol", [("dnapsw", 14, 11), ("bitstutter-noise", 4, 7), ("psw2dna", 4, 13)])
def test_host_traceback_equals_device_traceback(capi, machines, name, il, ol):
    """DPMatrix::traceBack with selectMaxTrans, walked on the host over the device-filled matrix, is the same path the
    device traceback kernel returns (both follow src/dpmatrix.defs.h:82-110 incl. the first-maximum tie-break)."""
    from machineboss_amd.dp import ViterbiMatrix, SeqPair
    preset = name != "bitstutter-noise"
    m, em = machines(name, None if preset else load_json("io", "params.json"), useDefaults=preset, preset=preset)
    x, y = synth_tokens(9, il, ol, em.nInTok, em.nOutTok)
    sp = SeqPair(em.inputTokenizer.detokenize(x), em.outputTokenizer.detokenize(y))
    vit = ViterbiMatrix(em, sp)
    dev, host = vit.path(m), vit.traceBack(m)
    assert dev.steps == host.steps and len(host.steps) > 0
    _check_path(em, m, host, x, y)


def test_sample_path_and_trace_forward(capi, machines):
    """ForwardMatrix::samplePath (src/forward.cpp:17-23) and traceForward over a Backward matrix (dpmatrix.defs.h:128-159):
    valid paths, deterministic in the mt19937 seed, sampled in proportion to exp(candidate)."""
    from machineboss_amd.dp import ForwardMatrix, BackwardMatrix, SeqPair, Mt19937, randomTransSelector
    m, em = machines("bitstutter-noise", load_json("io", "params.json"))
    sp = SeqPair(list("101"), list("10011"))
    fwd, back = ForwardMatrix(em, sp), BackwardMatrix(em, sp)
    x, y = fwd.input, fwd.output
    paths = [fwd.samplePath(m, Mt19937(seed, result_bits=32)) for seed in (1, 1, 2, 3, 4, 5)]
    for p in paths:
        _check_path(em, m, p, x, y)
    assert paths[0].steps == paths[1].steps
    # many alignments of similar weight under uniform parameters: different seeds must give different samples
    md, emd = machines("dnapsw", None, useDefaults=True, preset=True)
    xd, yd = synth_tokens(3, 8, 8, 4, 4)
    fd = ForwardMatrix(emd, SeqPair(emd.inputTokenizer.detokenize(xd), emd.outputTokenizer.detokenize(yd)))
    pd_ = [fd.samplePath(md, Mt19937(seed, result_bits=32)) for seed in range(1, 9)]      # 32-bit result_type: a real sampler (quirk Q12)
    for p in pd_:
        _check_path(emd, md, p, xd, yd)
    assert len({tuple(p.steps) for p in pd_}) > 1
    # forward walk over the Backward matrix from the start cell (the TraceTerminator overload honours its position)
    steps = []
    back.traceForwardFrom(m, 0, 0, 0, lambda ip, op, s, ti: steps.append((s, ti)) or False)
    from machineboss_amd.dp import MachinePath
    p = MachinePath([m.state[s].getTransition(ti) for s, ti in steps], steps)
    _check_path(em, m, p, x, y)


def test_backward_visitors_match_device_counts(capi, machines):
    """BackwardMatrix::getCounts visitor / postTransQueue / traceFrom (src/backward.cpp:52-108) on the host agree with
    the device count sweep (MachineCounts)."""
    from machineboss_amd.dp import ForwardMatrix, BackwardMatrix, MachineCounts, SeqPair
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    x, y = synth_tokens(21, 9, 12, em.nInTok, em.nOutTok)
    sp = SeqPair(em.inputTokenizer.detokenize(x), em.outputTokenizer.detokenize(y))
    fwd, back = ForwardMatrix(em, sp), BackwardMatrix(em, sp)
    dev = MachineCounts(em, [sp])
    host = MachineCounts(em)
    back.getCounts(fwd, host)
    assert close(host._flat, dev._flat, 1e-5, 1e-7)
    q = back.postTransQueue(fwd)
    assert all(q[k].weight >= q[k + 1].weight for k in range(len(q) - 1))
    per_edge = np.zeros(em.nTransitions)
    for pt in q:
        per_edge[em.transOffset[pt.src] + pt.transIndex] += pt.weight
    assert close(per_edge, host._flat, 1e-12, 1e-15)
    # path through the most probable transition usage: terminator overload visits a contiguous start->end path
    top = q[0]
    seen = []
    back.traceFrom(m, fwd, top.inPos, top.outPos, top.src, top.transIndex, lambda ip, op, s, ti: seen.append((ip, op, s, ti)) or False)
    assert (top.inPos, top.outPos, top.src, top.transIndex) in seen
    used = sorted(set((s, ti) for _, _, s, ti in seen))
    emitted_in = sum(1 for s, ti in [(a[2], a[3]) for a in seen] if m.state[s].getTransition(ti).inp)
    emitted_out = sum(1 for s, ti in [(a[2], a[3]) for a in seen] if m.state[s].getTransition(ti).out)
    assert emitted_in == len(x) and emitted_out == len(y) and len(used) > 0


@pytest.mark.parametrize("name,il,ol", [("dnapsw", 9, 12), ("bitstutter-noise", 3, 6), ("protpsw", 7, 5)])
def test_host_walkers_match_oracle_exactly(capi, oracle_mod, machines, name, il, ol):
    """The host walkers of dp.py (rows a7 / a10 / a13: samplePath, traceBack / traceForward with selectors and terminators,
    postTransQueue, traceFrom) against the oracle's restatements of src/dpmatrix.defs.h:61-186, src/forward.cpp:17-23 and
    src/backward.cpp:52-108, walking the SAME device-filled matrices with the same std::mt19937 stream: identical steps."""
    from machineboss_amd.dp import ForwardMatrix, BackwardMatrix, SeqPair, Mt19937, randomTransSelector
    preset = name != "bitstutter-noise"
    m, em = machines(name, None if preset else load_json("io", "params.json"), useDefaults=preset, preset=preset)
    om = oracle_mod.OracleMachine(em)
    x, y = synth_tokens(17, il, ol, em.nInTok, em.nOutTok)
    if name == "bitstutter-noise":
        x, y = em.inputTokenizer.tokenize(list("101")), em.outputTokenizer.tokenize(list("100110"))
    sp = SeqPair(em.inputTokenizer.detokenize(x), em.outputTokenizer.detokenize(y))
    fwd, back = ForwardMatrix(em, sp), BackwardMatrix(em, sp)
    F, B = fwd.cells(), back.cells()
    off = np.asarray(em.transOffset)
    eid = lambda steps: [int(off[s]) + int(ti) for s, ti in steps]
    # samplePath: several draws from one generator (the state of the stream carries over, as in stochasticDownsample), under
    # both readings of std::mt19937::result_type (quirk Q12: 64 bits with libstdc++ on Linux, 32 with libc++)
    for bits in (64, 32):
        oracle_mod.set_result_bits(bits)
        try:
            g_host, g_or = Mt19937(42, result_bits=bits), oracle_mod.Mt19937(42)
            drawn = []
            for _ in range(4):
                drawn.append(eid(fwd.samplePath(m, g_host).steps))
                assert drawn[-1] == list(om.trace_back(x, y, F, rng=g_or)[::-1])
        finally:
            oracle_mod.set_result_bits(64)
    # traceBack with a terminator from an interior cell that carries probability, max and random selectors
    for (ip, op) in [(len(x), len(y)), (len(x) // 2, len(y) // 2)]:
        for s in range(em.nStates):
            if not F[op, ip, s] > -math.inf:
                continue
            steps = []
            fwd.traceBackFrom(m, ip, op, s, lambda a, b, src, ti: steps.append((src, ti)) or False)
            assert eid(steps) == list(om.trace_back(x, y, F, ip, op, s))
            steps = []; gh, go = Mt19937(7 + s), oracle_mod.Mt19937(7 + s)      # (default reading: 64-bit result_type)
            fwd.traceBackFrom(m, ip, op, s, lambda a, b, src, ti: steps.append((src, ti)) or False, randomTransSelector(gh))
            assert eid(steps) == list(om.trace_back(x, y, F, ip, op, s, rng=go))
            if B[op, ip, s] > -math.inf:
                steps = []
                back.traceForwardFrom(m, ip, op, s, lambda a, b, src, ti: steps.append((src, ti)) or False)
                assert eid(steps) == list(om.trace_forward(x, y, B, ip, op, s))
    # postTransQueue: the same usages with the same weights, largest first
    q = back.postTransQueue(fwd)
    ipo, opo, eo, wo = om.post_trans(x, y, F, B)
    assert len(q) == len(wo)
    order = np.argsort(-wo, kind="stable")
    assert [pt.weight for pt in q] == [wo[k] for k in order]
    assert [(pt.inPos, pt.outPos, int(off[pt.src]) + pt.transIndex) for pt in q] == [(int(ipo[k]), int(opo[k]), int(eo[k])) for k in order]
    # Machine::downsample's loop (src/machine.cpp:2053-2076): traceFrom with its terminator, over the head of the queue
    allowed = np.zeros(em.nTransitions, np.uint8); mask = np.zeros(em.nTransitions, np.uint8)

    def stop(a, b, src, ti):
        e = int(off[src]) + ti
        if allowed[e]:
            return True
        allowed[e] = 1
        return False
    # (PostTrans carries the transition's DESTINATION cell, src/backward.cpp:77-83, and traceFrom treats it as the source's: off
    #  the empty pair Machine::downsample uses, such a trace can step onto a -inf cell -- an Assert in the reference, an error
    #  here and in the oracle, at the same step and with the same transitions marked so far)
    from machineboss_amd.machine import MachineError
    for pt in q[:10]:
        failed = [False, False]
        try:
            back.traceFrom(m, fwd, pt.inPos, pt.outPos, pt.src, pt.transIndex, stop)
        except MachineError:
            failed[0] = True
        try:
            om.trace_from(x, y, F, B, pt.inPos, pt.outPos, int(off[pt.src]) + pt.transIndex, mask)
        except RuntimeError:
            failed[1] = True
        assert failed[0] == failed[1] and np.array_equal(allowed, mask)
    assert allowed.sum() > 0


# ---- envelopes (src/seqpair.h:75-97; DPMatrix fills visit only the cells inside, dpmatrix.h:142-144 reads -inf outside) ----
@pytest.mark.parametrize("name,il,ol,width", [("dnapsw", 14, 17, 2), ("dnapsw", 150, 170, 5), ("bitstutter-noise", 4, 6, 1), ("psw2dna", 3, 11, 1),
                                              ("psw2dna", 40, 130, 4)])
def test_envelope_fills_match_oracle(capi, oracle_mod, machines, name, il, ol, width):
    """Forward / Viterbi / Backward / counts under a path-area envelope: device vs oracle."""
    from machineboss_amd.seqpair import Envelope
    preset = name != "bitstutter-noise"
    m, em = machines(name, None if preset else load_json("io", "params.json"), useDefaults=preset, preset=preset)
    om = oracle_mod.OracleMachine(em)
    dm = capi.DeviceMachine(em)
    x, y = synth_tokens(31, il, ol, em.nInTok, em.nOutTok)
    # an alignment that matches min(il, ol) symbols diagonally, then gaps -> its path-area envelope of the given width
    k = min(il, ol)
    cols = [("a", "b")] * k + [("a", "")] * (il - k) + [("", "b")] * (ol - k)
    if name == "psw2dna" and 3 * il <= ol:      # protein against DNA: one residue per codon, so that the band holds real alignments
        cols = [("a", "b"), ("", "b"), ("", "b")] * il + [("", "b")] * (ol - 3 * il)
    env = Envelope.pathAreaEnvelope(cols, width)
    assert env.connected() and not env.isFull()
    with oracle_mod.envelope(env.inStart, env.inEnd):
        Vr = om.viterbi(x, y); Fr = om.forward(x, y, oracle_mod.SUM_EXACT); Br = om.backward(x, y, oracle_mod.SUM_EXACT)
        cr = np.zeros(em.nTransitions); llr = om.counts_add(x, y, cr, oracle_mod.SUM_EXACT)
        path_r = om.traceback(x, y, Vr) if Vr[-1, -1, -1] > -math.inf else None
    V = dm.fill(capi.MB_VITERBI, x, y, 0, env.inStart, env.inEnd)
    F = dm.fill(capi.MB_FORWARD, x, y, 0, env.inStart, env.inEnd)
    B = dm.fill(capi.MB_BACKWARD, x, y, 0, env.inStart, env.inEnd)
    # envelopes stay on the fast families: machines of <= 16 states on the small-machine family, larger ones on the tiled
    # family (cells outside are forced to -inf inside the sweep, tiles outside the envelope are not launched)
    small = em.nStates <= 16
    assert capi.last_kernel_name().startswith("k_small_") if small else capi.last_kernel_name() == "k_medium_jit"
    rel, abs_ = FAST_REL, FAST_ABS
    assert np.array_equal(V, Vr) and close(F, Fr, rel, abs_) and close(B, Br, rel, abs_)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y), (x, y)])
    b.set_envelopes([(env.inStart, env.inEnd), None])          # second pair keeps the full envelope
    ll = b.forward(capi.MB_ROLLING)
    assert close(ll[0], Fr[-1, -1, -1], max(rel, 1e-10), abs_) and close(ll[1], om.loglike(x, y, oracle_mod.SUM_EXACT), max(rel, 1e-10), abs_)
    vll, off, edges = b.viterbi()
    assert vll[0] == Vr[-1, -1, -1]
    if path_r is not None:
        assert np.array_equal(edges[off[0]:off[1]], path_r)
    counts, s, cll = b.counts()
    cf = np.zeros(em.nTransitions); llf = om.counts_add(x, y, cf, oracle_mod.SUM_EXACT)
    if math.isfinite(llr):
        assert close(counts, cr + cf, 1e-5, 1e-7) and close(s, llr + llf, max(rel, 1e-10), abs_)
    else:                                        # the band holds no alignment: that pair counts nothing (the reference would produce NaN)
        assert close(counts, cf, 1e-5, 1e-7) and s == -math.inf
    # errors of DPMatrix::alloc (src/dpmatrix.defs.h:31-32)
    with pytest.raises(capi.MbError, match="mismatch"):
        b.set_envelopes([(env.inStart[:-1], env.inEnd[:-1]), None])
    gap = list(env.inStart); gap[1] = env.inEnd[0] + 2
    if gap[1] < il:
        with pytest.raises(capi.MbError, match="not connected|mismatch"):
            b.set_envelopes([(gap, [max(a + 1, e) for a, e in zip(gap, env.inEnd)]), None])


def test_path_envelope_through_dp_classes(capi, machines):
    """Quirk Q1: an aligned SeqPair is filled under its alignment's PATH envelope (src/dpmatrix.defs.h:16-17,
    seqpair.cpp:104-110), so MachineCounts over t/io/pathlist.json counts exactly the aligned transitions."""
    from machineboss_amd.dp import ForwardMatrix, MachineCounts
    from machineboss_amd.seqpair import seqPairListFromJson
    p = load_json("io", "params.json")
    m, em = machines("bitnoise", p)
    aligned = seqPairListFromJson(load_json("io", "pathlist.json"))
    plain = seqPairListFromJson(load_json("io", "seqpairlist.json"))
    fa, fp = ForwardMatrix(em, aligned[0]), ForwardMatrix(em, plain[0])
    assert not fa.env.isFull() and fp.env.isFull() and fa.logLike() <= fp.logLike() and fa.cell(0, 1, 0) == -math.inf
    mc = MachineCounts(em, aligned)
    # 3 + 2 aligned columns, each a match transition of the single state; nothing else is reachable in the envelope
    assert abs(sum(sum(r) for r in mc.count) - 5.0) < 1e-9
    # the batch front ends keep the same envelope: --loglike of an aligned pair is its path's weight, and ViterbiMatrix::path
    # traces inside the envelope the matrix was filled in (it agrees with logLike() and with the host walk over that matrix)
    from machineboss_amd.dp import ViterbiMatrix, forwardLogLikeBatch
    assert forwardLogLikeBatch(em, [aligned[0], plain[0]]) == [fa.logLike(), fp.logLike()]
    va = ViterbiMatrix(em, aligned[0])
    dev, host = va.path(m), va.traceBack(m)
    assert dev.steps == host.steps and len(dev.steps) == 3
    assert abs(sum(em.logWeight[em.transOffset[s] + ti] for s, ti in dev.steps) - va.logLike()) < 1e-12


@pytest.mark.parametrize("idx", range(10))
def test_reference_js_tier_goldens_through_gpu(capi, idx, monkeypatch):
    """Outputs of the reference's own JavaScript CPU implementation (tests/golden/js/, generated by running
    js/webgpu/cpu/*-2d.mjs with node in the dev container): Viterbi score bit for bit, Forward / Backward log-likelihood
    within 1e-4 relative (observed ~1e-9) through the HIP path."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    cases = {c["name"]: c for c in load_json("js", "cases.json")}
    gold = load_json("js", "goldens.json")[idx]
    case = cases[gold["name"]]
    m = Machine.fromFile(golden_path(*case["machine"].split("/")))
    defs = m.getParamDefs(True); defs.update(case["params"])
    em = EvaluatedMachine.fromMachine(m, defs)
    x = np.array(case["input"], np.int32); y = np.array(case["output"], np.int32)
    one = gold.get("oneTape")
    # cases 5-9: one-tape machines of the reference's 1-D tier (js/webgpu/cpu/*-1d.mjs) through the ONE-TAPE family -- the sweep generated
    # for the machine and the interpreter, one workgroup per sequence and cut for k workgroups
    variants = [{}] if not one else [{"MB_WIDE_MIN_STATES": "1"}, {"MB_WIDE_MIN_STATES": "1", "MB_WIDE_JIT": "0"},
                                      {"MB_WIDE_MIN_STATES": "1", "MB_ONETAPE_PARTS_MIN_LEN": "0", "MB_ONETAPE_PARTS": "3"}]
    for knobs in variants:
        for k_, v_ in knobs.items(): monkeypatch.setenv(k_, v_)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        for flags in (capi.MB_MATERIALISE, capi.MB_ROLLING):
            ll = b.forward(flags)[0]
            assert abs(ll - float(gold["forward"])) <= 1e-4 * abs(float(gold["forward"])) and abs(ll - float(gold["forward"])) < (1e-6 if not one else FAST_REL * abs(float(gold["forward"])) + FAST_ABS)
            if one: assert "k_wide" in capi.last_kernel_name() or "k_onetape" in capi.last_kernel_name(), capi.last_kernel_name()
        assert b.viterbi(paths=False)[0][0] == float(gold["viterbi"])
        B = dm.fill(capi.MB_BACKWARD, x, y)
        assert abs(B[0, 0, 0] - float(gold["backward"])) < (1e-6 if not one else FAST_REL * abs(float(gold["backward"])) + FAST_ABS)
        if one and "forwardCells" in gold:      # every cell of the reference's grids, [position][state]
            F = dm.fill(capi.MB_FORWARD, x, y)
            S = em.nStates
            fr = np.array([-np.inf if v == "-inf" else float(v) for v in gold["forwardCells"]]).reshape(-1, S)
            br = np.array([-np.inf if v == "-inf" else float(v) for v in gold["backwardCells"]]).reshape(-1, S)
            Fm = F[:, 0, :] if one == "out" else F[0, :, :]
            Bm = B[:, 0, :] if one == "out" else B[0, :, :]
            assert close(Fm, fr, FAST_REL, FAST_ABS) and close(Bm, br, FAST_REL, FAST_ABS)
        for k_ in knobs: monkeypatch.delenv(k_)
        dm.close()


# ---- M-step and Baum-Welch (src/counts.cpp:117-295, src/fitter.cpp) around the device count sweep -------------------------
def _round4(d):
    return {k: float("%.4g" % v) for k, v in d.items()}


def test_mstep_reference_golden(capi, machines):
    """t/src/testmaximize.cpp on bitnoise (Makefile:502): counts -> MachineObjective::optimize == {"p":0.6667,"q":0.3333}."""
    from machineboss_amd.dp import MachineCounts
    from machineboss_amd.fitter import MachineObjective
    from machineboss_amd.machine import Constraints
    from machineboss_amd.seqpair import SeqPair
    params = load_json("io", "params.json")
    m, em = machines("bitnoise", params)
    counts = MachineCounts(em, [SeqPair.fromJson(load_json("io", "tiny.json"))])
    cons = Constraints.fromJson(load_json("io", "pqcons.json"))
    obj = MachineObjective(m, counts, cons, {})
    opt = obj.optimize(params)
    assert _round4(opt) == load_json("expect", "max-bitnoise-params-tiny.json")
    # the general path (the reference's transformed parameterisation minimised with BFGS) reaches the same optimum
    obj._closed_form = lambda seed: None
    assert _round4(obj.optimize(params)) == load_json("expect", "max-bitnoise-params-tiny.json")


@pytest.mark.parametrize("data", ["seqpairlist.json", "pathlist.json"])
def test_fit_reference_golden(capi, machines, data):
    """`boss t/machine/bitnoise.json -N t/io/pqcons.json -D <data> -T` (Makefile:505-507): plain pairs and aligned pairs
    (path envelopes) both fit to {"p":0.4,"q":0.6}."""
    from machineboss_amd.fitter import MachineFitter
    from machineboss_amd.machine import Constraints, Machine
    from machineboss_amd.seqpair import seqPairListFromJson
    m = Machine.fromFile(golden_path("machine", "bitnoise.json"))
    fitter = MachineFitter(m, Constraints.fromJson(load_json("io", "pqcons.json")))
    fit = fitter.fit(seqPairListFromJson(load_json("io", data)))
    assert _round4(fit) == load_json("expect", "fit-bitnoise-seqpairlist.json")
    assert len(fitter.log) >= 2 and all(b >= a - 1e-9 for a, b in zip(fitter.log, fitter.log[1:]))   # EM never decreases the likelihood


def test_fit_protpsw_recovers_likelihood(capi, machines):
    """EM on protpsw (BASELINE config 3 in miniature): the likelihood rises monotonically and the fitted parameters
    stay normalised; weights are reloaded with mb_machine_set_weights between iterations."""
    from machineboss_amd.fitter import MachineFitter
    from machineboss_amd.machine import Machine
    from machineboss_amd.seqpair import SeqPair
    m = Machine.fromFile(golden_path("preset", "protpsw.json"))
    em0 = machines("protpsw", None, useDefaults=True, preset=True)[1]
    rng = np.random.RandomState(5)
    pairs = []
    for k in range(6):
        x = rng.randint(1, 21, size=12)
        y = x.copy(); y[rng.randint(0, 12, size=3)] = rng.randint(1, 21, size=3)    # a noisy copy: substitutions only
        pairs.append(SeqPair(em0.inputTokenizer.detokenize(x), em0.outputTokenizer.detokenize(y)))
    fitter = MachineFitter(m)
    fit = fitter.fit(pairs)
    assert len(fitter.log) >= 3 and all(b >= a - 1e-7 for a, b in zip(fitter.log, fitter.log[1:]))
    for g in m.cons.norm:
        assert abs(sum(fit[p] for p in g) - 1.0) < 1e-9
    assert fitter.log[-1] > fitter.log[0] + 1.0


# ---- boss-compatible command line (target/boss.cpp:716-847) around the batched device calls --------------------------------
def _boss(argv):
    import io
    from machineboss_amd import boss
    buf = io.StringIO()
    assert boss.run(argv, out=buf) == 0
    return buf.getvalue()


def test_boss_cli_reference_outputs(capi):
    """CLI-level expectations of the reference (Makefile:493-531,567-572): -L / -V text, --align JSON incl. meta.path,
    --counts parameter counts, --train fits."""
    mach = lambda n: golden_path("machine", n + ".json")
    io_ = lambda n: golden_path("io", n + ".json")
    exp = lambda n: open(golden_path("expect", n + ".json")).read()
    # test-101-bitnoise-001: boss bitnoise -P params --input-chars 101 --output-chars 001 -L
    got = json.loads(_boss([mach("bitnoise"), "-P", io_("params"), "--input-chars", "101", "--output-chars", "001", "-L"]))
    assert got[0][:2] == ["101", "001"] and float("%.4g" % got[0][2]) == json.loads(exp("101-bitnoise-001"))[0][0]   # expectation is name-stripped, 4 digits
    got = json.loads(_boss([mach("bitstutter-noise"), "-P", io_("params"), "--input-chars", "101", "--output-chars", "0011", "-V"]))
    assert got[0][:2] == ["101", "0011"] and float("%.3g" % got[0][2]) == json.loads(exp("101-bitstutternoise-vit-0011"))[0][0]
    got = json.loads(_boss([mach("bitstutter-noise"), "-P", io_("params"), "--input-chars", "101", "--output-chars", "0011", "-L"]))
    assert float("%.3g" % got[0][2]) == json.loads(exp("101-bitstutternoise-fwd-0011"))[0][0]
    # test-align-stutter-noise: --align prints the SeqPairList with alignment and meta.path, byte for byte
    got = _boss([mach("bitstutter-noise"), "-P", io_("params"), "-D", io_("difflen"), "-A"])
    assert got == exp("align-stutter-noise-difflen")
    # test-counts: -C prints parameter counts {"p":2,"q":1}
    got = _boss([mach("bitnoise"), "-P", io_("params"), "--input-chars", "101", "--output-chars", "001", "-C"])
    assert json.loads(got) == json.loads(exp("counts"))
    # test-counts (Makefile:518): the same counts from the composed-with-sequences form -- generator(101) . bitnoise .
    # recognizer(001) is an all-silent machine whose only evidence is the empty pair
    got = _boss(["--generate-chars", "101", mach("bitnoise"), "--recognize-chars", "001", "-P", io_("params"), "-N", io_("pqcons"), "-C"])
    assert json.loads(got) == json.loads(exp("counts"))
    # test-counts3 (Makefile:524): a generator with a counting parameter
    got = _boss([mach("counter"), "--output-chars", "xxx", "-C"])
    assert json.loads(got) == json.loads(exp("counter"))
    # test-fit-bitnoise-seqpairlist: -T prints the fitted parameters
    got = json.loads(_boss([mach("bitnoise"), "-N", io_("pqcons"), "-D", io_("seqpairlist"), "-T"]))
    assert {k: float("%.4g" % v) for k, v in got.items()} == json.loads(exp("fit-bitnoise-seqpairlist"))
    # a pair the machine cannot tokenise prints "-Infinity" (target/boss.cpp:797-805)
    got = _boss([mach("bitnoise"), "-P", io_("params"), "--input-chars", "10x", "--output-chars", "001", "-L"])
    assert '"-Infinity"' in got
    # presets and --use-defaults: the SURVEY anchor value of protpsw 50x50 is reproduced to the 6 digits boss prints
    a = load_json("survey_anchors.json")["anchors"][0]
    m = __import__("machineboss_amd.machine", fromlist=["Machine"]).Machine.fromFile(golden_path("preset", "protpsw.json"))
    em = __import__("machineboss_amd.evalmachine", fromlist=["EvaluatedMachine"]).EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    x, y = synth_tokens(a["seed"], a["inLen"], a["outLen"], em.nInTok, em.nOutTok)
    got = json.loads(_boss(["--preset", "protpsw", "--use-defaults", "--input-chars", "".join(em.inputTokenizer.detokenize(x)),
                            "--output-chars", "".join(em.outputTokenizer.detokenize(y)), "-L"]))
    assert got[0][2] == float("%.6g" % a["forward"])


# ---- BASELINE.json configurations at FULL size: size-independent properties (the oracle would need hours) -----------------
def _count_invariants(em, counts, nPairs, inLen, outLen):
    """Expected usage of input-consuming transitions sums to the number of input symbols (likewise output): every
    path of a pair reads each symbol exactly once."""
    cin = counts[np.asarray(em.inTok) != 0].sum(); cout = counts[np.asarray(em.outTok) != 0].sum()
    # posterior = exp(F + w + B - logLike): the fp32 correction terms of the tiled fills leave ~1e-6 absolute error in
    # F + B - logLike after a few thousand steps, i.e. ~1e-6..5e-6 relative in every count (measured 2e-6 at 1 kb x 1 kb)
    tol = 2e-5
    return abs(cin - nPairs * inLen) <= tol * nPairs * inLen and abs(cout - nPairs * outLen) <= tol * nPairs * outLen


@pytest.mark.parametrize("preset,config,nPairs,il,ol", [("dnapsw", 2, 1024, 1000, 1000), ("protpsw", 3, 1024, 400, 400),
                                                       ("psw2dna", 4, 256, 487, 10000)])
def test_baseline_configs_full_size_properties(capi, machines, preset, config, nPairs, il, ol):
    """configs[1] (dnapsw, 1024 x 1 kb), configs[2] per GPU (protpsw, 1024 x 400 aa) and configs[3] (psw2dna, 10 kb DNA):
    rolling == materialised Forward; Viterbi <= Forward; duplicated pairs give identical results; the Viterbi path
    re-scores to the Viterbi log-likelihood and spells both sequences; counts obey the symbol-count invariant and the
    summed Forward log-likelihood equals the Forward pass's."""
    m, em = machines(preset, None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    inTok, inOff, outTok, outOff = synth_batch(config, nPairs, il, ol, em.nInTok, em.nOutTok)
    # make pair 1 a copy of pair 0: identical inputs must give bit-identical outputs wherever they run
    inTok[inOff[1]:inOff[2]] = inTok[inOff[0]:inOff[1]]; outTok[outOff[1]:outOff[2]] = outTok[outOff[0]:outOff[1]]
    b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
    assert b.cells() == nPairs * (il + 1) * (ol + 1) * em.nStates
    llm = b.forward(capi.MB_MATERIALISE); llr = b.forward(capi.MB_ROLLING)
    assert capi.last_kernel_name() == ("k_medium_jit" if preset == "psw2dna" else "k_small_sum_roll")
    assert np.all(np.isfinite(llm)) and close(llr, llm, 1e-12) and llm[0] == llm[1] and llr[0] == llr[1]
    # tracebacks: all 256 pairs of config 4 (one traceback byte per cell: 340 GB of bytes in two chunks), a sub-batch for the others
    npv = nPairs if preset == "psw2dna" else min(nPairs, 64)
    bv = capi.DeviceBatch(dm, inTok[:inOff[npv]], inOff[:npv + 1], outTok[:outOff[npv]], outOff[:npv + 1])
    vll, off, edges = bv.viterbi()
    assert np.all(vll <= llm[:npv] + 1e-9) and vll[0] == vll[1]
    lw = np.asarray(em.logWeight)
    for k in (0, 1, npv - 1):
        e = edges[off[k]:off[k + 1]]
        # path is contiguous start -> end, spells the pair, and its weight IS the Viterbi score (summed in path order)
        assert em.src[e[0]] == 0 and em.dst[e[-1]] == em.nStates - 1 and np.array_equal(em.dst[e[:-1]], em.src[e[1:]])
        assert np.array_equal(em.inTok[e][em.inTok[e] != 0], inTok[inOff[k]:inOff[k + 1]])
        assert np.array_equal(em.outTok[e][em.outTok[e] != 0], outTok[outOff[k]:outOff[k + 1]])
        acc = 0.0
        for w in lw[e]:
            acc += w
        assert abs(acc - vll[k]) <= 1e-9 * abs(vll[k])
    nc = min(nPairs, 256 if preset != "psw2dna" else 24)    # config 4: more pairs than one chunk of Backward matrices holds (21)
    bc = capi.DeviceBatch(dm, inTok[:inOff[nc]], inOff[:nc + 1], outTok[:outOff[nc]], outOff[:nc + 1])
    counts, s, cll = bc.counts()
    assert _count_invariants(em, counts, nc, il, ol)
    assert close(cll, llm[:nc], 1e-8) and abs(s - cll.sum()) <= 1e-9 * abs(s)


# ---- machines assembled by composition (algebra.py = Machine::compose) through the DP engine -------------------------------
def test_composed_machines_through_gpu(capi, oracle_mod):
    """protpsw . translate (177 states, composed here) and the cons-stripped three-way protpsw . translate . dnapsw
    (482 states, BASELINE config 4b): Viterbi bit-exact, Forward/Backward within the tiled tolerance, tracebacks and
    counts vs the oracle -- and the reference's `boss bitstutter.json bitnoise.json ... -A` CLI golden."""
    from machineboss_amd import algebra as A
    from machineboss_amd.machine import Constraints, Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    pt = A.compose(P("protpsw"), P("translate"))
    d = P("dnapsw"); d.cons = Constraints()
    for mach, il, ol in ((pt, 7, 33), (A.compose(pt, d), 5, 18)):
        em = EvaluatedMachine.fromMachine(mach, None, useDefaults=True)
        om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
        x, y = synth_tokens(12, il, ol, em.nInTok, 3)      # DNA over {A,C,G}: no stop codons, which translate cannot emit
        V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
        assert "k_medium" in capi.last_kernel_name() and V[-1, -1, -1] > -math.inf
        assert np.array_equal(V, om.viterbi(x, y))
        assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        assert close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        vll, off, edges = b.viterbi()
        assert np.array_equal(edges, om.traceback(x, y, om.viterbi(x, y)))
        counts, s, _ = b.counts()
        ref = np.zeros(em.nTransitions); om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
        assert close(counts, ref, 1e-5, 1e-7)
    got = _boss([golden_path("machine", "bitstutter.json"), golden_path("machine", "bitnoise.json"), "-P", golden_path("io", "params.json"),
                 "-D", golden_path("io", "difflen.json"), "-A"])
    assert got == open(golden_path("expect", "align-stutter-noise-difflen.json")).read()


def test_rolling_small_batch_uses_pipeline(capi, oracle_mod, machines, monkeypatch):
    """Fewer pairs than CUs: MB_ROLLING is served by the tile pipeline (same log-likelihoods, only they are kept)."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(60 + k, 20 + k, 70, em.nInTok, em.nOutTok) for k in range(3)]
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    ref = b.forward(capi.MB_ROLLING); n0 = capi.last_launch_count()
    monkeypatch.setenv("MB_ROLLING_MIN_PAIRS", "192")
    got = b.forward(capi.MB_ROLLING); n1 = capi.last_launch_count()
    assert n1 != n0 and close(got, ref, 1e-12)
    monkeypatch.setenv("MB_MEDIUM_ROLLTILES", "0")       # round 2: the same tiles WITH the matrix stored for nothing
    assert close(b.forward(capi.MB_ROLLING), got, 1e-12)
    monkeypatch.delenv("MB_MEDIUM_ROLLTILES")
    pairs2 = pairs + [synth_tokens(70, 150, 700, em.nInTok, em.nOutTok)]     # several strips and blocks: halo columns and boundary records
    b2 = capi.DeviceBatch.from_pairs(dm, pairs2)
    g2 = b2.forward(capi.MB_ROLLING)
    assert capi.last_kernel_name() == "k_medium_jit" and close(g2, b2.forward(capi.MB_MATERIALISE), 1e-12)
    om = oracle_mod.OracleMachine(em)
    assert close(got, [om.loglike(x, y, oracle_mod.SUM_EXACT) for x, y in pairs], FAST_REL, FAST_ABS)


def test_memory_budget_chunking(capi, machines):
    """A batch whose matrices exceed the device-memory budget is processed in sub-batches / recycled matrix slots with
    identical results; a single matrix that cannot fit is refused with the library's message."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(300 + k, 30 + (k % 5), 200 + 7 * k, em.nInTok, em.nOutTok) for k in range(12)]
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    ref_ll = b.forward(capi.MB_MATERIALISE); ref_v = b.viterbi(); ref_c = b.counts()
    one = max((len(x) + 1) * (len(y) + 1) * em.nStates * 8 for x, y in pairs)
    try:
        capi.release_workspace()
        capi.set_memory_budget(int(2.5 * one))           # two matrices at most: Forward recycles slots, counts runs pair by pair
        assert np.array_equal(b.forward(capi.MB_MATERIALISE), ref_ll)
        v = b.viterbi()
        assert np.array_equal(v[0], ref_v[0]) and np.array_equal(v[1], ref_v[1]) and np.array_equal(v[2], ref_v[2])
        c = b.counts()
        assert close(c[0], ref_c[0], 1e-5, 1e-9) and np.array_equal(c[2], ref_c[2])   # per-tile fp32 usage sums: tiling differs per sub-batch
        capi.set_memory_budget(one // 2)                  # no fp64 matrix fits: a materialised fill is refused ...
        with pytest.raises(capi.MbError, match="exceeds the device memory budget"):
            dm.fill(capi.MB_FORWARD, *pairs[-1])
        v = b.viterbi()                                   # ... while Viterbi (one traceback byte per cell, an eighth of it) still runs
        assert np.array_equal(v[0], ref_v[0]) and np.array_equal(v[2], ref_v[2])
        capi.set_memory_budget(one // 16)
        with pytest.raises(capi.MbError, match="exceeds the device memory budget"):
            b.viterbi()
    finally:
        capi.set_memory_budget(0)
        capi.release_workspace()


def test_one_tape_machines(capi, oracle_mod):
    """Generators (no input alphabet) and recognisers (no output alphabet): the lattice degenerates to one row / column;
    silent chains and a 1-state machine included (BASELINE config 5 is a one-tape machine)."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    gen = Machine.fromJson({"state": [
        {"id": "S", "trans": [{"to": "A"}, {"to": "B", "weight": 0.25}]},
        {"id": "A", "trans": [{"to": "A", "out": "x", "weight": 0.5}, {"to": "B", "out": "y", "weight": 0.3}, {"to": "E", "weight": 0.2}]},
        {"id": "B", "trans": [{"to": "A", "out": "y", "weight": 0.6}, {"to": "B", "out": "x", "weight": 0.1}, {"to": "E", "weight": 0.3}]},
        {"id": "E"}]})
    rec = Machine.fromJson({"state": [{"id": "S", "trans": [{"to": "S", "in": "a", "weight": 0.4}, {"to": "S", "in": "b", "weight": 0.1}]}]})
    for mach, il, ol in ((gen, 0, 23), (rec, 17, 0), (gen, 0, 0)):
        em = EvaluatedMachine.fromMachine(mach, {})
        om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
        rng = np.random.RandomState(il + ol)
        x = rng.randint(1, max(em.nInTok, 1) + 1, size=il).astype(np.int32) if em.nInTok else np.zeros(0, np.int32)
        y = rng.randint(1, max(em.nOutTok, 1) + 1, size=ol).astype(np.int32) if em.nOutTok else np.zeros(0, np.int32)
        for fam in (capi.KERNEL_GENERIC, capi.KERNEL_AUTO):
            capi.set_kernel(fam)
            try:
                V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
                b = capi.DeviceBatch.from_pairs(dm, [(x, y)] * 3)
                ll = b.forward(capi.MB_ROLLING); vll, off, edges = b.viterbi(); counts, s, _ = b.counts()
            finally:
                capi.set_kernel(capi.KERNEL_AUTO)
            assert np.array_equal(V, om.viterbi(x, y))
            assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
            assert close(ll, [F[-1, -1, -1]] * 3, 1e-9, 1e-12)
            if V[-1, -1, -1] > -math.inf:
                assert np.array_equal(edges[off[0]:off[1]], om.traceback(x, y, om.viterbi(x, y)))
            ref = np.zeros(em.nTransitions); om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
            assert close(counts, 3 * ref, 1e-5, 1e-7)


@pytest.mark.parametrize("stages,split", [(0, 12), (1, 12), (2, 2), (3, 3), (5, 2), (9, 1)])
def test_closure_stages_and_node_splitting(capi, oracle_mod, machines, monkeypatch, stages, split):
    """Every shape of the program compiler -- levelled, full closure, K-stage closures, with nodes split into parts of
    every granularity -- against the oracle: psw2dna (emitting and silent fan-out of 62) and random machines with match
    edges, in both sweep directions; Viterbi stays bit-exact whatever the split."""
    from randmachine import random_machine, random_seq
    monkeypatch.setenv("MB_MEDIUM_CLOSURE_STAGES", str(stages))
    monkeypatch.setenv("MB_MEDIUM_SPLIT_DEGREE", str(split))
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    rng = np.random.RandomState(stages * 7 + split)
    cases = [(em, *synth_tokens(50 + stages, 37, 61, em.nInTok, em.nOutTok))]
    for S, seed in ((29, 4), (75, 5)):
        rm = random_machine(S, 3, 4, seed, allow_inf=(seed == 5))
        cases.append((rm, random_seq(rng, 23, 3), random_seq(rng, 31, 4)))
    for e, x, y in cases:
        om = oracle_mod.OracleMachine(e); dm = capi.DeviceMachine(e)
        V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
        assert "k_medium" in capi.last_kernel_name()
        assert np.array_equal(V, om.viterbi(x, y))
        assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        assert close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        assert close(b.forward(capi.MB_ROLLING), [F[-1, -1, -1]], 1e-9, 1e-12)
        counts, s, _ = b.counts()
        if F[-1, -1, -1] > -math.inf:      # an unreachable pair counts nothing here (the reference would produce NaN)
            ref = np.zeros(e.nTransitions); om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
            assert close(counts, ref, 1e-5, 1e-7)
        else:
            assert not counts.any()


def _profile_machine(nodes):
    """fn3 profile truncated to `nodes` nodes . simple_introns . translate . dnapsw (BASELINE config 5, literal composition)."""
    from machineboss_amd import algebra as A
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(nodes)
    m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    return m, EvaluatedMachine.fromMachine(m, None, useDefaults=True)


@pytest.mark.parametrize("stages", [None, 0, 1, 3, 40, -2, -999])
def test_one_tape_family_profile_machine(capi, oracle_mod, monkeypatch, stages):
    """BASELINE config 5 at test size: a 3-node profile composed with simple_introns . translate . dnapsw (762 states,
    one tape) through the one-tape kernel family -- Forward / Backward matrices, rolling log-likelihood, bit-exact Viterbi
    matrix and path, posterior counts -- for the levelled program and several closure groupings, LDS and L2 vectors."""
    m, em = _profile_machine(3)
    assert em.nInTok == 0 and em.nStates >= 256
    monkeypatch.setenv("MB_WIDE_RETIMED", "0")        # the column-by-column kernels (the retimed sweep: test_one_tape_retimed_sweep)
    if stages is not None:                              # None: the planner's own choice; -2: adaptive stages of <= 2 slots; -999: stage cuts by dynamic programming
        monkeypatch.setenv("MB_WIDE_CLOSURE_STAGES", str(stages))
    if stages == 3:
        monkeypatch.setenv("MB_WIDE_GLOBAL_VECTORS", "1")
    if stages in (0, 1):
        monkeypatch.setenv("MB_WIDE_FAST_INDEX", "0")
    if stages in (1, 40):
        monkeypatch.setenv("MB_WIDE_HYBRID", "1")     # ... with the previous column in L2 (closure programs only)
    if stages in (None, 1, 40):
        monkeypatch.setenv("MB_WIDE_FP32", "1")       # the fp32-relative log-sum-exp kernel (default only when fp64 columns exceed the LDS)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(11 + abs(stages or 0))
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32) for n in (41, 0, 1, 17)]
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    ll = b.forward(capi.MB_ROLLING)
    assert capi.last_kernel_name().startswith("k_wide_")
    llm = b.forward(capi.MB_MATERIALISE); vll, off, edges = b.viterbi(); counts, s, _ = b.counts()
    assert capi.last_kernel_name() == "k_onetape_counts"
    ref = np.zeros(em.nTransitions)
    for k, y in enumerate(ys):
        V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
        Vo = om.viterbi(x, y)
        assert np.array_equal(V, Vo)
        assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        assert close([ll[k], llm[k]], [F[-1, -1, -1]] * 2, 1e-9, 1e-12) and vll[k] == Vo[-1, -1, -1]
        if Vo[-1, -1, -1] > -math.inf:
            assert np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, Vo))
        if F[-1, -1, -1] > -math.inf:
            om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
    assert close(counts, ref, 1e-5, 1e-7)
    # a weight update rebuilds the programs
    lw = np.array(em.logWeight, dtype=np.float64) - 0.125
    dm.set_weights(lw); om.set_weights(lw)
    assert close(b.forward(capi.MB_ROLLING), [om.loglike(x, y, oracle_mod.SUM_EXACT) for y in ys], FAST_REL, FAST_ABS)


@pytest.mark.parametrize("nSym,S", [(70, 300), (100, 2600), (70, 10400)])
def test_one_tape_machines_the_retimed_planner_declines(capi, oracle_mod, nSym, S):
    """Why the column-by-column sweeps (k_wide_sweep, k_wide_viterbi, k_wide_sum32) stay in the library: the retimed planner keeps one penalty column per
    symbol in a 64-entry row (mb_wide.hip, wide_ret_build: rowLen > 64 declines), so a one-tape machine over more than 63 symbols -- a
    byte or codon-pair alphabet; the reference puts no bound on an alphabet (src/machine.h:95-101) -- gets no retimed program.  No knob
    is set here: the family's own choice, checked like every other one-tape kernel.  10 400 states: two fp64 columns no longer fit the LDS
    and the sum semiring takes the fp32-relative kernel (k_wide_sum32), also by itself.  The max semiring: up to 2 048 states the tile
    kernel specialised at run time (mb_api.hip, onetape_tiled_viterbi), k_wide_viterbi while a column fits the LDS, k_wide_sweep<1> beyond."""
    from randmachine import random_machine
    em = random_machine(S, 0, nSym, 9100 + nSym, density=2.2, silent_density=1.0)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(nSym)
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, nSym + 1, size=n).astype(np.int32) for n in (37, 0, 1, 90)]
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    ll = b.forward(capi.MB_ROLLING)
    f32 = capi.last_kernel_name() == "k_wide_sum32"
    assert f32 == (S > 10000) and (f32 or capi.last_kernel_name() == "k_wide_sweep<0>")
    rel, ab = (2e-5, 2e-5) if f32 else (FAST_REL, FAST_ABS)
    b.viterbi(paths=False)
    assert capi.last_kernel_name() == ("k_medium_jit" if S <= 2048 else ("k_wide_viterbi" if S < 10000 else "k_wide_sweep<1>"))
    vll, off, edges = b.viterbi()
    counts, s, _ = b.counts()
    ref = np.zeros(em.nTransitions)
    for k, y in enumerate(ys):
        V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y)
        Vo = om.viterbi(x, y)
        assert np.array_equal(V, Vo) and vll[k] == Vo[-1, -1, -1]
        assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), rel, ab) and close(ll[k], F[-1, -1, -1], 1e-6 if f32 else 1e-9, 1e-12)
        if Vo[-1, -1, -1] > -math.inf:
            assert np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, Vo))
        if F[-1, -1, -1] > -math.inf:
            om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
    assert close(counts, ref, 1e-4 if f32 else 1e-5, 1e-6 if f32 else 1e-7)


@pytest.mark.parametrize("knobs", [{}, {"MB_WIDE_RETIMED_PERIOD": "+3"}, {"MB_WIDE_LANES": "256"}, {"MB_WIDE_GLOBAL_VECTORS": "1"}, {"MB_ONETAPE_TB": "0"}, {"MB_ONETAPE_TB": "0", "MB_WIDE_GLOBAL_VECTORS": "1"}, {"MB_ONETAPE_TB_FAST": "0"}])
def test_one_tape_retimed_sweep(capi, oracle_mod, monkeypatch, knobs):
    """The retimed sweep of the one-tape family (mb_wide.hip k_wide_retimed: every state on its own column, a period of a
    few wide rounds instead of one round per silent level): Viterbi matrices bit for bit, Forward / Backward matrices and
    rolling log-likelihoods within the fast-path tolerance, paths and counts -- tiny generators and recognisers (period 1),
    the fn3 profile (protein alphabet: 22 penalty columns, 44 columns in flight, relay entries), the 3-node composite of
    config 5 (period 9); lengths on both sides of the 64-column token window; a longer period than the shortest,
    256-lane workgroups (several rounds per residue), and the ring in an L2-resident vector instead of the LDS (what
    machines of tens of thousands of states get)."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    monkeypatch.setenv("MB_WIDE_MIN_STATES", "1")
    monkeypatch.setenv("MB_ONETAPE_TRACEBACK_MIN_TRANS", "0")      # ... and the one-tape traceback walker (k_onetape_traceback) on every machine
    for k, v in knobs.items():
        if k != "MB_WIDE_RETIMED_PERIOD": monkeypatch.setenv(k, v)
    gen = Machine.fromJson({"state": [
        {"id": "S", "trans": [{"to": "A"}, {"to": "B", "weight": 0.25}]},
        {"id": "A", "trans": [{"to": "A", "out": "x", "weight": 0.5}, {"to": "B", "out": "y", "weight": 0.3}, {"to": "E", "weight": 0.2}]},
        {"id": "B", "trans": [{"to": "A", "out": "y", "weight": 0.6}, {"to": "B", "out": "x", "weight": 0.1}, {"to": "E", "weight": 0.3}]},
        {"id": "E"}]})
    rec = Machine.fromJson(json.loads(json.dumps({"state": [{"id": st.name, "trans": [dict(to=t.dest, weight=t.weight, **({"in": t.out} if t.out else {})) for t in st.trans]} for st in gen.state]})))
    cases = [(EvaluatedMachine.fromMachine(gen, {}), 1, 1, (0, 1, 23, 150)), (EvaluatedMachine.fromMachine(rec, {}), 0, 1, (0, 5, 70)),
             (EvaluatedMachine.fromMachine(HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).machine(True), {}), 1, 2, (0, 1, 40, 130)),
             (_profile_machine(3)[1], 1, 9, (0, 1, 41, 100))]
    for em, tape, pMin, lens in cases:
        if "MB_WIDE_RETIMED_PERIOD" in knobs: monkeypatch.setenv("MB_WIDE_RETIMED_PERIOD", str(pMin + int(knobs["MB_WIDE_RETIMED_PERIOD"])))
        om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
        nt = em.nOutTok if tape else em.nInTok
        z = np.zeros(0, np.int32)
        seqs = [np.random.RandomState(100 + n).randint(1, nt + 1, size=n).astype(np.int32) for n in lens]
        pairs = [(z, q) if tape else (q, z) for q in seqs]
        ref = np.zeros(em.nTransitions)
        for x, y in pairs:
            V = dm.fill(capi.MB_VITERBI, x, y); assert _one_wg(capi.last_kernel_name()) == ("k_wide_retimed<1,L2>" if "MB_WIDE_GLOBAL_VECTORS" in knobs else "k_wide_retimed<1>")
            F = dm.fill(capi.MB_FORWARD, x, y); assert _kn(capi.last_kernel_name()).startswith("k_wide_retimed<0")
            B = dm.fill(capi.MB_BACKWARD, x, y)
            assert np.array_equal(V, om.viterbi(x, y))
            assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
            if F[-1, -1, -1] > -math.inf: om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        ll = b.forward(capi.MB_ROLLING); assert _kn(capi.last_kernel_name()).startswith("k_wide_retimed<0")
        vll, off, edges = b.viterbi()
        # (paths: one traceback CODE per cell kept by the max sweep, walked by k_onetape_traceback_codes -- round 4, the default since the
        # code sweep's reduction became two butterflies -- or, MB_ONETAPE_TB=0, the fp64 matrix and its walkers)
        assert ("codes" in capi.last_kernel_name()) == (knobs.get("MB_ONETAPE_TB") != "0")
        counts, s, _ = b.counts()
        assert close(counts, ref, 1e-5, 1e-7)
        for k, (x, y) in enumerate(pairs):
            Vo = om.viterbi(x, y)
            assert close([ll[k]], [om.loglike(x, y, oracle_mod.SUM_EXACT)], FAST_REL, FAST_ABS) and vll[k] == Vo[-1, -1, -1]
            if Vo[-1, -1, -1] > -math.inf:
                assert np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, Vo))
        # a weight update rebuilds the program (same schedule, new records)
        lw = np.array(em.logWeight, dtype=np.float64) - 0.0625
        dm.set_weights(lw); om.set_weights(lw)
        x, y = pairs[-1]
        assert np.array_equal(dm.fill(capi.MB_VITERBI, x, y), om.viterbi(x, y))
        dm.close()


def test_one_tape_retimed_single_stage_period(capi, oracle_mod, tmp_path):
    """All-emitting one-tape machines (plain HMMs) have a retimed period of ONE stage, and when its slot count is a multiple
    of the kernel's prefetch ring the period's only barrier sits in its last slot: the next period's token penalties, written
    at the top of the period, are then looked up before that barrier -- round 3 relied on timing there.
    Matrices of every semiring against the oracle, several sequences so that all CUs of an XCD are busy at once."""
    from randmachine import plain_hmm
    hit = 0
    for S, fan in ((600, 9), (1024, 11), (512, 11), (300, 4)):      # (the first three: barrier in the last slot; the fourth: padding behind it)
        em = plain_hmm(S, fan, 4, 7 + S)
        h = capi.debug_wide_retimed(em, str(tmp_path / "ret.bin"), mode=capi.MB_VITERBI)
        assert h["period"] == 1
        hit += bool((h["records"][0, -1, :]["pad"] & 0x40000000).any())      # the barrier slot is the last slot of the period
        om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
        z = np.zeros(0, np.int32)
        ys = [np.random.RandomState(300 + n).randint(1, 5, size=n).astype(np.int32) for n in (1, 9, 70, 200)]
        for y in ys:
            V = dm.fill(capi.MB_VITERBI, z, y); assert _kn(capi.last_kernel_name()).startswith("k_wide_retimed<1")
            assert np.array_equal(V, om.viterbi(z, y))
            assert close(dm.fill(capi.MB_FORWARD, z, y), om.forward(z, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
            assert close(dm.fill(capi.MB_BACKWARD, z, y), om.backward(z, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        b = capi.DeviceBatch.from_pairs(dm, [(z, ys[k % 4]) for k in range(96)])
        ll = b.forward(capi.MB_ROLLING); vll = b.viterbi(paths=False)[0]
        for k in range(96):
            assert close([ll[k]], [om.loglike(z, ys[k % 4], oracle_mod.SUM_EXACT)], FAST_REL, FAST_ABS)
            assert vll[k] == om.viterbi(z, ys[k % 4])[-1, -1, -1]
        dm.close()
    assert hit == 3


def test_one_tape_count_kernel(capi, oracle_mod, monkeypatch):
    """Posterior counts of one-tape machines (lane = transition, mb_wide.hip k_onetape_counts): long sequences cut into
    column parts, many short ones, a recogniser (the tape is the input), after a weight update; against the per-cell kernel
    on the same matrices and against the oracle."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    m, em = _profile_machine(3)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(77)
    x = np.zeros(0, np.int32)
    for lens in ([700, 513, 64], [9, 0, 31, 2] * 10):
        ys = [rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32) for n in lens]
        b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
        c1, s1, l1 = b.counts()
        assert capi.last_kernel_name() == "k_onetape_counts"
        monkeypatch.setenv("MB_ONETAPE_COUNTS", "0")
        c0, s0, l0 = b.counts()
        monkeypatch.delenv("MB_ONETAPE_COUNTS")
        assert capi.last_kernel_name() != "k_onetape_counts"
        # (the per-cell kernel divides by the likelihood as the reference does; the lane-per-transition kernels divide every column's
        # terms by what its emitting terms sum to -- round 4, k_onetape_counts_lds -- which removes the rounding F and B collect along
        # 700 columns: the two agree to that drift, and the normalised one is the closer to the oracle)
        assert close(c1, c0, 2e-5, 1e-9) and s1 == s0 and np.array_equal(l1, l0)
        refc = np.zeros(em.nTransitions)
        for y in ys: om.counts_add(x, y, refc, oracle_mod.SUM_EXACT)
        assert close(c1, refc, 1e-5, 1e-7) and close(c0, refc, 1e-5, 1e-7)
        # every output symbol is emitted by exactly one transition of a path: expected emissions sum to the symbol count
        assert abs(c1[np.asarray(em.outTok) != 0].sum() - sum(lens)) <= 1e-6 * max(1, sum(lens))
    lw = np.array(em.logWeight, dtype=np.float64) - 0.0625 * (np.arange(em.nTransitions) % 3)
    dm.set_weights(lw); om.set_weights(lw)
    y = rng.randint(1, em.nOutTok + 1, size=37).astype(np.int32)
    ref = np.zeros(em.nTransitions); om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
    c, _, _ = capi.DeviceBatch.from_pairs(dm, [(x, y)]).counts()
    assert capi.last_kernel_name() == "k_onetape_counts" and close(c, ref, 1e-5, 1e-7)
    # a recogniser with enough states for the one-tape family: a ring of 300 states reading a/b
    states = [{"id": "s%d" % k, "trans": [{"to": "s%d" % ((k + 1) % 300), "in": "ab"[k % 2], "weight": 0.5},
                                          {"to": "s%d" % ((k + 7) % 300), "in": "ab"[(k + 1) % 2], "weight": 0.25}] +
               ([{"to": "end", "weight": 0.25}] if k % 5 == 0 else [])} for k in range(300)] + [{"id": "end"}]
    er = EvaluatedMachine.fromMachine(Machine.fromJson({"state": states}), {})
    orr = oracle_mod.OracleMachine(er); dr = capi.DeviceMachine(er)
    xs = [rng.randint(1, 3, size=n).astype(np.int32) for n in (200, 45, 0, 130)]
    br = capi.DeviceBatch.from_pairs(dr, [(xx, np.zeros(0, np.int32)) for xx in xs])
    cr, sr, llr = br.counts()
    assert capi.last_kernel_name() == "k_onetape_counts"
    ref = np.zeros(er.nTransitions)
    for xx, l in zip(xs, llr):
        if l > -math.inf:
            orr.counts_add(xx, np.zeros(0, np.int32), ref, oracle_mod.SUM_EXACT)
    assert ref.any() and close(cr, ref, 1e-5, 1e-7)


def test_one_tape_split_forward(capi, oracle_mod, monkeypatch):
    """Few sequences on a one-tape machine: the log-likelihood-only Forward cuts every sequence in two (Forward over the prefix,
    Backward over the suffix, side by side; k_onetape_join sums over the emitting transitions that cross the cut) -- against
    the plain sweep, the materialised fill and the oracle; generator and recogniser; fp64 and fp32-relative kernels."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    m, em = _profile_machine(3)
    om = oracle_mod.OracleMachine(em)
    rng = np.random.RandomState(5)
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32) for n in (64, 151, 400, 65)]
    ref = [om.loglike(x, y, oracle_mod.SUM_EXACT) for y in ys[:2]]
    for fp32 in ("0", "1"):
        monkeypatch.setenv("MB_WIDE_FP32", fp32)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
        ll = b.forward(capi.MB_ROLLING)
        assert "k_onetape_join" in capi.last_kernel_name()
        monkeypatch.setenv("MB_ONETAPE_SPLIT", "0")
        ll0 = b.forward(capi.MB_ROLLING)
        assert "join" not in capi.last_kernel_name()
        monkeypatch.delenv("MB_ONETAPE_SPLIT")
        llm = b.forward(capi.MB_MATERIALISE)
        assert close(ll, ll0, FAST_REL, FAST_ABS) and close(ll, llm, FAST_REL, FAST_ABS)   # the fp32 correction terms fall differently in the two halves
        assert close(ll[:2], ref, FAST_REL, FAST_ABS)
        # a sequence shorter than the threshold keeps the whole batch on the plain sweep
        b2 = capi.DeviceBatch.from_pairs(dm, [(x, ys[0]), (x, ys[0][:10])])
        b2.forward(capi.MB_ROLLING)
        assert "join" not in capi.last_kernel_name()
    monkeypatch.delenv("MB_WIDE_FP32")
    # recogniser: a ring of 300 states reading a/b, exits every fifth state
    states = [{"id": "s%d" % k, "trans": [{"to": "s%d" % ((k + 1) % 300), "in": "ab"[k % 2], "weight": 0.5},
                                          {"to": "s%d" % ((k + 7) % 300), "in": "ab"[(k + 1) % 2], "weight": 0.25}] +
               ([{"to": "end", "weight": 0.25}] if k % 5 == 0 else [])} for k in range(300)] + [{"id": "end"}]
    er = EvaluatedMachine.fromMachine(Machine.fromJson({"state": states}), {})
    orr = oracle_mod.OracleMachine(er); dr = capi.DeviceMachine(er)
    xs = [rng.randint(1, 3, size=n).astype(np.int32) for n in (200, 75, 130)]
    br = capi.DeviceBatch.from_pairs(dr, [(xx, np.zeros(0, np.int32)) for xx in xs])
    llr = br.forward(capi.MB_ROLLING)
    assert "k_onetape_join" in capi.last_kernel_name()
    assert close(llr, [orr.loglike(xx, np.zeros(0, np.int32), oracle_mod.SUM_EXACT) for xx in xs], FAST_REL, FAST_ABS)


@pytest.mark.parametrize("knobs", [{}, {"MB_ONETAPE_PARTS": "7", "MB_ONETAPE_PART_LANES": "128"}, {"MB_ONETAPE_PARTS": "2"}, {"MB_ONETAPE_PARTS": "3", "MB_ONETAPE_PART_LANES": "1024", "MB_ONETAPE_PART_EXCLUSIVE": "0"}])
def test_one_tape_k_workgroups_per_sequence(capi, oracle_mod, monkeypatch, knobs):
    """k workgroups per sequence (k_wide_retimed_parts): with fewer sequences than CUs the machine is cut into parts along
    a topological order of its strongly connected components, one retimed program and one workgroup per part, values crossing through
    an exchange buffer.  Against the ONE-workgroup sweep (MB_ONETAPE_PARTS=1) and the oracle, on config 5's literal composition with a
    3-node profile (762 states) and ragged lengths (an empty sequence among them): Viterbi matrices, scores and paths bit for bit --
    through fp64 cells and through traceback codes --, Forward / Backward matrices to 1e-8 relative (the same sums in another order:
    other lane groups, two-transition candidates), rolling log-likelihoods (cut in two on top) and counts within the fast-path tolerance of the oracle;
    default cut, seven parts of two wavefronts, two parts, three parts sharing CUs."""
    m, em = _profile_machine(3)
    om = oracle_mod.OracleMachine(em)
    rng = np.random.RandomState(11)
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32) for n in (0, 1, 63, 200, 129, 64)]
    pairs = [(x, y) for y in ys]
    monkeypatch.setenv("MB_WIDE_MIN_STATES", "1")
    monkeypatch.setenv("MB_ONETAPE_TRACEBACK_MIN_TRANS", "0")
    monkeypatch.setenv("MB_ONETAPE_PARTS_MIN_LEN", "0")      # (by default a machine is cut for sweeps of thousands of columns only: the planning costs 0.5-1 s)
    def run(dm):
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        out = {"names": []}
        def note(): out["names"].append(capi.last_kernel_name())
        out["roll"] = b.forward(capi.MB_ROLLING); note()
        out["mat"] = b.forward(capi.MB_MATERIALISE); note()
        out["vit"] = b.viterbi(); note()
        monkeypatch.setenv("MB_ONETAPE_TB", "0")
        out["vit64"] = b.viterbi(); note()
        monkeypatch.delenv("MB_ONETAPE_TB")
        out["cnt"] = b.counts(); note()
        y = ys[3]
        out["V"] = dm.fill(capi.MB_VITERBI, x, y); note()
        out["F"] = dm.fill(capi.MB_FORWARD, x, y); note()
        out["B"] = dm.fill(capi.MB_BACKWARD, x, y); note()
        return out
    monkeypatch.setenv("MB_ONETAPE_PARTS", "1")
    dm1 = capi.DeviceMachine(em)
    one = run(dm1)
    assert not any("parts" in n for n in one["names"])
    monkeypatch.delenv("MB_ONETAPE_PARTS")
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    dmk = capi.DeviceMachine(em)
    got = run(dmk)
    want = int(knobs.get("MB_ONETAPE_PARTS", "8"))      # (the default cap of a machine whose ring fits one CU)
    sweeps = [n for n in got["names"] if "k_wide_retimed" in n or "k_wide_jit" in n]      # (k_wide_jit: the sweep generated for this machine and this cut, mb_wide_jit.cpp)
    assert sweeps and all(" parts" in n for n in sweeps if not (want == 2 and "<1" in n and "MB_ONETAPE_PARTS" not in knobs)), got["names"]
    assert any("in %d parts" % want in n for n in sweeps) or want == 7, got["names"]      # (a cut may come out with fewer parts than asked for)
    # bit for bit against the one-workgroup sweep
    assert np.array_equal(got["V"], one["V"])
    for key in ("F", "B", "mat"):      # (other lane groups: another order of the same sum, and other fp32 roundings of its correction term)
        d = np.abs(np.asarray(got[key]) - np.asarray(one[key])); d = d[np.isfinite(d)]
        assert np.array_equal(np.isneginf(got[key]), np.isneginf(one[key])) and (d.size == 0 or d.max() < 5e-6), (key, float(d.max()))      # (cells of -400: 1e-8 relative)
    for key in ("vit", "vit64"):
        for a, b_ in zip(got[key], one[key]): assert np.array_equal(np.asarray(a), np.asarray(b_)), key
    assert close(got["roll"], one["roll"], 1e-6, 1e-6) and close(got["cnt"][0], one["cnt"][0], 1e-5, 1e-8)      # (the fp32 correction terms fall differently)
    # ... and against the oracle
    y = ys[3]
    assert np.array_equal(got["V"], om.viterbi(x, y))
    assert close(got["F"], om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(got["B"], om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
    vll, off, edges = got["vit"]
    ref = np.zeros(em.nTransitions)
    for k, (xx, yy) in enumerate(pairs):
        Vo = om.viterbi(xx, yy)
        assert vll[k] == Vo[-1, -1, -1] and close([got["roll"][k]], [om.loglike(xx, yy, oracle_mod.SUM_EXACT)], FAST_REL, FAST_ABS)
        if Vo[-1, -1, -1] > -math.inf:
            assert np.array_equal(edges[off[k]:off[k + 1]], om.traceback(xx, yy, Vo))
            om.counts_add(xx, yy, ref, oracle_mod.SUM_EXACT)
    assert close(got["cnt"][0], ref, 1e-5, 1e-7)
    # a weight update rebuilds the parts
    lw = np.array(em.logWeight, dtype=np.float64) - 0.0625
    dmk.set_weights(lw); om.set_weights(lw)
    assert np.array_equal(dmk.fill(capi.MB_VITERBI, x, y), om.viterbi(x, y)) and " parts" in capi.last_kernel_name()
    dm1.close(); dmk.close()


def test_one_tape_machines_are_cut_for_long_sweeps_only(capi, monkeypatch):
    """Cutting a machine and planning its parts costs 0.5-1 s: by default only a launch whose longest sequence has >= 4 096 symbols runs k
    workgroups per sequence (MB_ONETAPE_PARTS_MIN_LEN), and MB_ONETAPE_PARTS=1 turns the cut off; the results agree either way."""
    m, em = _profile_machine(3)
    x = np.zeros(0, np.int32)
    rng = np.random.RandomState(17)
    short = [(x, rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32)) for n in (300, 120)]
    long_ = [(x, rng.randint(1, em.nOutTok + 1, size=n).astype(np.int32)) for n in (4500, 120)]
    dm = capi.DeviceMachine(em)
    bs, bl = capi.DeviceBatch.from_pairs(dm, short), capi.DeviceBatch.from_pairs(dm, long_)
    vs = bs.viterbi(paths=False)[0]; assert " parts" not in capi.last_kernel_name()
    vl = bl.viterbi(paths=False)[0]; assert " parts" in capi.last_kernel_name()
    monkeypatch.setenv("MB_ONETAPE_PARTS", "1")
    assert np.array_equal(bl.viterbi(paths=False)[0], vl) and " parts" not in capi.last_kernel_name()
    monkeypatch.delenv("MB_ONETAPE_PARTS")
    monkeypatch.setenv("MB_ONETAPE_PARTS_MIN_LEN", "100")
    assert np.array_equal(bs.viterbi(paths=False)[0], vs) and " parts" in capi.last_kernel_name()
    dm.close()


def test_one_tape_parts_follow_a_weight_update_that_changes_the_edge_set(capi, oracle_mod, monkeypatch):
    """the two-transition candidates of a part are remembered as indices into the part's edge list, and that list drops
    edges of weight -inf -- a weight update that silences one transition and revives another keeps the list's LENGTH but not its
    edges.  The remembered choice is keyed by a signature of the list now: after such an update the k-part sweep still returns what a
    fresh machine with those weights returns, and what the oracle says."""
    import copy
    m, em = _profile_machine(3)
    x = np.zeros(0, np.int32)
    ys = [np.random.RandomState(30 + n).randint(1, 4, size=n).astype(np.int32) for n in (90, 150)]
    monkeypatch.setenv("MB_ONETAPE_PARTS_MIN_LEN", "0")
    monkeypatch.setenv("MB_WIDE_MIN_STATES", "1")
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    b.viterbi(paths=False); assert " parts" in capi.last_kernel_name()
    lw = np.asarray(em.logWeight).copy()
    sil = np.nonzero((np.asarray(em.inTok) == 0) & (np.asarray(em.outTok) == 0) & np.isfinite(lw))[0]
    rng = np.random.RandomState(4)
    for trial in range(3):
        lw2 = lw.copy()
        kill = rng.choice(sil, size=3, replace=False)
        lw2[kill] = -np.inf                                   # three silent transitions silenced ...
        dm.set_weights(lw2)
        got1 = b.viterbi(paths=False)[0]; ll1 = b.forward(capi.MB_ROLLING)
        lw3 = lw2.copy(); lw3[kill[0]] = lw[kill[0]]; lw3[rng.choice(np.setdiff1d(sil, kill))] = -np.inf      # ... then one revived and another silenced: same count
        dm.set_weights(lw3)
        got = b.viterbi(paths=False)[0]; ll = b.forward(capi.MB_ROLLING)
        assert " parts" in capi.last_kernel_name() or "k_onetape_join" in capi.last_kernel_name()
        em3 = copy.copy(em); em3.logWeight = lw3
        dmf = capi.DeviceMachine(em3)
        bf = capi.DeviceBatch.from_pairs(dmf, [(x, y) for y in ys])
        assert np.array_equal(got, bf.viterbi(paths=False)[0]) and close(ll, bf.forward(capi.MB_ROLLING), 1e-9, 1e-9)
        om = oracle_mod.OracleMachine(em3)
        for k, y in enumerate(ys):
            assert got[k] == om.viterbi(x, y)[-1, -1, -1] and close(ll[k], om.loglike(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        dmf.close()
    dm.close()


def test_one_tape_parts_fall_back_when_a_value_never_arrives(capi, monkeypatch, capfd):
    """The exchange between the parts of a sequence has no way to hang: a lane waits a bounded time for another part's value, then raises
    the launch's status word, every other waiter stops, the kernel drains -- and the host discards the results, LATCHES the machine's
    program to one workgroup per sequence and runs the call once more (a shared device or a CU mask must cost one time-out, not
    a failure of every call).  Provoked with a time-out of a microsecond (a consumer's first value takes longer than that to arrive): the
    call succeeds with the one-workgroup sweep's results, a warning goes to stderr, and a fresh machine object cuts again."""
    m, em = _profile_machine(3)
    x = np.zeros(0, np.int32)
    ys = [np.random.RandomState(3 + n).randint(1, em.nOutTok + 1, size=n).astype(np.int32) for n in (80, 120)]
    monkeypatch.setenv("MB_ONETAPE_PARTS_MIN_LEN", "0")
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    good = b.viterbi(paths=False)[0]
    assert " parts" in capi.last_kernel_name()
    goodll = b.forward(capi.MB_ROLLING)
    for jit in ("1", "0"):      # the generated kernel and the interpreter
        monkeypatch.setenv("MB_WIDE_JIT", jit)
        monkeypatch.setenv("MB_ONETAPE_PART_TIMEOUT_S", "0.000001")
        dm2 = capi.DeviceMachine(em)
        b2 = capi.DeviceBatch.from_pairs(dm2, [(x, y) for y in ys])
        t0 = time.perf_counter()
        got = b2.viterbi(paths=False)[0]
        assert time.perf_counter() - t0 < 30
        assert np.array_equal(got, good) and " parts" not in capi.last_kernel_name()      # (latched: one workgroup per sequence)
        assert "waited longer" in capfd.readouterr().err
        ll = b2.forward(capi.MB_ROLLING)      # the Forward / Backward programs of the cut-in-two sweep time out and fall back as well
        assert close(ll, goodll, 1e-6, 1e-6)
        monkeypatch.delenv("MB_ONETAPE_PART_TIMEOUT_S")
        assert np.array_equal(b2.viterbi(paths=False)[0], good) and " parts" not in capi.last_kernel_name()      # still latched
        dm3 = capi.DeviceMachine(em)      # a new machine object: new programs, cut again
        b3 = capi.DeviceBatch.from_pairs(dm3, [(x, y) for y in ys])
        assert np.array_equal(b3.viterbi(paths=False)[0], good) and " parts" in capi.last_kernel_name()
        dm2.close(); dm3.close()
    dm.close()


@pytest.mark.parametrize("nodes,nSeq,L", [(20, 64, 2000), (86, 8, 300)])
def test_baseline_config5_full_size_properties(capi, oracle_mod, monkeypatch, nodes, nSeq, L):
    """BASELINE config 5 at the sizes that select the one-tape family's DEFAULT paths (no environment forcing): the 20-node
    machine (5 063 states: retimed sweep, ring in LDS) on 64 x 2 kb and the whole fn3 profile (21 761 states: retimed sweep,
    ring in L2) on 8 x 300 nt.  Rolling == materialised; the other arithmetic variant agrees to 1e-6; the Viterbi
    path re-scores to the Viterbi score and spells the sequence; counts keep the symbol-count invariant; one sequence of
    30 nt against the oracle."""
    m, em = _profile_machine(nodes)
    assert em.nInTok == 0 and em.nStates == {20: 5063, 86: 21761}[nodes]
    dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(nodes)
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, 4, size=L).astype(np.int32) for _ in range(nSeq)]      # DNA over {A,C,G}: no stop codons
    ys[1] = ys[0].copy()
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    llr = b.forward(capi.MB_ROLLING)
    kern = capi.last_kernel_name()
    import re
    assert re.match(r"k_wide_(retimed|jit)<0>", kern) and (nodes == 20 or re.match(r"k_wide_(retimed|jit)<0> in \d+ parts", kern))      # (21 761 states: one workgroup keeps its ring in L2, the parts of eight sequences keep theirs in LDS)   # (+ " x2 + k_onetape_join": few sequences are cut in two)
    nm = min(nSeq, 4)                                                            # matrices of a few sequences (21 761 x 301 doubles each)
    bm = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys[:nm]])
    llm = bm.forward(capi.MB_MATERIALISE)
    assert np.all(np.isfinite(llr)) and close(llm, llr[:nm], FAST_REL, FAST_ABS) and llr[0] == llr[1]
    # the other arithmetic (fp32-relative <-> fp64 columns) on a fresh machine object
    other = ("MB_WIDE_FP32", "1") if nodes == 20 else ("MB_WIDE_RETIMED", "0")   # (the whole profile without the retimed sweep: fp32-relative columns, the previous one in L2)
    monkeypatch.setenv(*other)
    dm2 = capi.DeviceMachine(em)
    b2 = capi.DeviceBatch.from_pairs(dm2, [(x, y) for y in ys[:nm]])
    ll2 = b2.forward(capi.MB_ROLLING)
    assert capi.last_kernel_name().startswith("k_wide_sum32") and close(ll2, llr[:nm], 1e-6)
    monkeypatch.delenv(other[0])
    # Viterbi: score <= Forward, the path is contiguous, spells the sequence and re-scores to the score
    bv = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys[:2]])
    vll, off, edges = bv.viterbi()
    lw = np.asarray(em.logWeight)
    assert np.all(vll <= llr[:2] + 1e-9) and vll[0] == vll[1]
    e = edges[off[0]:off[1]]
    assert em.src[e[0]] == 0 and em.dst[e[-1]] == em.nStates - 1 and np.array_equal(em.dst[e[:-1]], em.src[e[1:]])
    assert np.array_equal(em.outTok[e][em.outTok[e] != 0], ys[0])
    acc = 0.0
    for w in lw[e]:
        acc += w
    assert abs(acc - vll[0]) <= 1e-9 * abs(vll[0])
    # counts: every output symbol is emitted exactly once per sequence
    counts, s, cll = bv.counts()
    assert abs(counts[np.asarray(em.outTok) != 0].sum() - 2 * L) <= 1e-6 * 2 * L and close(cll, llr[:2], 1e-8)
    # one short sequence against the oracle
    om = oracle_mod.OracleMachine(em)
    y30 = ys[0][:30]
    b30 = capi.DeviceBatch.from_pairs(dm, [(x, y30)])
    ref = om.loglike(x, y30, oracle_mod.SUM_EXACT)
    assert close(b30.forward(capi.MB_ROLLING), [ref], FAST_REL, FAST_ABS)
    v30, o30, e30 = b30.viterbi()
    V = om.viterbi(x, y30)
    assert v30[0] == V[-1, -1, -1] and np.array_equal(e30, om.traceback(x, y30, V))


def test_baseline_config5_at_its_stated_length(capi, monkeypatch):
    """BASELINE config 5 as stated -- 64 sequences x 50 kb on the ~5k-state machine (20-node fn3 profile . simple_introns .
    translate . dnapsw, 5 063 states) -- on one GPU, through size-independent properties (the oracle needs minutes per
    sequence here): the default log-likelihood path (every sequence CUT IN TWO, Forward over the prefix || Backward over the
    suffix, joined over the crossing transitions, 25 000 columns per half) == the plain rolling sweep == the materialised
    fill of a sub-batch; the fp32-relative arithmetic (running reference R over 50 000 columns) agrees to 1e-6; the Viterbi
    path of a 50 kb sequence is contiguous, spells it and re-scores to the Viterbi score; the count sweep emits every symbol
    exactly once; identical sequences give identical results wherever they run."""
    L, nSeq = 50000, 64
    m, em = _profile_machine(20)
    assert em.nInTok == 0 and em.nStates == 5063
    dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(2050)
    x = np.zeros(0, np.int32)
    ys = [rng.randint(1, 4, size=L).astype(np.int32) for _ in range(nSeq)]      # DNA over {A,C,G}: no stop codons
    ys[1] = ys[0].copy(); ys[63] = ys[0].copy()
    b = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys])
    assert b.cells() == nSeq * (L + 1) * 5063
    llr = b.forward(capi.MB_ROLLING)
    assert _one_wg(capi.last_kernel_name()) == "k_wide_retimed<0> x2 + k_onetape_join"     # 2 x 64 workgroups <= 256 CUs: cut in two
    assert np.all(np.isfinite(llr)) and llr[0] == llr[1] == llr[63]
    monkeypatch.setenv("MB_ONETAPE_SPLIT", "0")
    llp = b.forward(capi.MB_ROLLING)
    assert _one_wg(capi.last_kernel_name()) == "k_wide_retimed<0>" and close(llr, llp, 1e-8) and llp[0] == llp[1] == llp[63]
    monkeypatch.delenv("MB_ONETAPE_SPLIT")
    b2 = capi.DeviceBatch.from_pairs(dm, [(x, y) for y in ys[:2]])                # 2 x 2.03 GB matrices
    llm = b2.forward(capi.MB_MATERIALISE)
    assert close(llm, llp[:2], 1e-9, 1e-12)                                       # the plain rolling sweep is the materialised one without the stores
    monkeypatch.setenv("MB_WIDE_FP32", "1")
    dm32 = capi.DeviceMachine(em)
    b32 = capi.DeviceBatch.from_pairs(dm32, [(x, y) for y in ys[:4]])
    monkeypatch.setenv("MB_ONETAPE_SPLIT", "0")
    ll32 = b32.forward(capi.MB_ROLLING)
    assert capi.last_kernel_name().startswith("k_wide_sum32") and close(ll32, llp[:4], 1e-6)
    monkeypatch.delenv("MB_ONETAPE_SPLIT"); monkeypatch.delenv("MB_WIDE_FP32")
    dm32.close()
    vll, off, edges = b2.viterbi()
    lw = np.asarray(em.logWeight)
    assert np.all(vll <= llp[:2] + 1e-9) and vll[0] == vll[1] and np.array_equal(edges[off[0]:off[1]], edges[off[1]:off[2]])
    e = edges[off[0]:off[1]]
    assert em.src[e[0]] == 0 and em.dst[e[-1]] == em.nStates - 1 and np.array_equal(em.dst[e[:-1]], em.src[e[1:]])
    assert np.array_equal(em.outTok[e][em.outTok[e] != 0], ys[0])
    acc = 0.0
    for w in lw[e]:
        acc += w
    assert abs(acc - vll[0]) <= 1e-9 * abs(vll[0])
    # every symbol is emitted exactly once (round 3: to 2e-4 at this length -- the rounding of 50 000 columns of log-sum-exp in F and B;
    # round 4 divides every column's terms by what its emitting terms sum to, k_onetape_counts_lds)
    counts, s, cll = b2.counts()
    # (the E-step's fills carry their correction term in fp64 at this length: their log-likelihood is the exact one to 1e-14, the rolling
    #  sweep's fp32 term is 1.1e-8 away from it -- test_baseline_config5_one_sequence_at_50kb_against_the_oracle has both against the oracle)
    assert abs(counts[np.asarray(em.outTok) != 0].sum() - 2 * L) <= 1e-6 * 2 * L and close(cll, llp[:2], 5e-8)


def test_pipelined_forward_matches_plain(capi, machines):
    """Config 4a's 256-pair pipeline (matrix slots recycled while the sweep is in flight, medium_forward_pipelined) against
    the same pairs filled without recycling: identical log-likelihoods."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    inTok, inOff, outTok, outOff = synth_batch(4, 24, 487, 1500, em.nInTok, em.nOutTok)
    b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
    plain = b.forward(capi.MB_MATERIALISE)
    one = (487 + 1) * (1500 + 1) * em.nStates * 8
    capi.set_memory_budget(5 * one)            # room for five matrices: the other 19 pairs reuse their slots
    try:
        piped = b.forward(capi.MB_MATERIALISE)
        launches = capi.last_launch_count()
    finally:
        capi.set_memory_budget(0)
    assert np.array_equal(piped, plain) and launches > 0 and "k_medium" in capi.last_kernel_name()


def test_one_tape_family_small_machines(capi, oracle_mod, monkeypatch):
    """The one-tape family forced onto tiny generators (256-lane workgroups, single lanes per state)."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    monkeypatch.setenv("MB_WIDE_MIN_STATES", "1")
    gen = Machine.fromJson({"state": [
        {"id": "S", "trans": [{"to": "A"}, {"to": "B", "weight": 0.25}]},
        {"id": "A", "trans": [{"to": "A", "out": "x", "weight": 0.5}, {"to": "B", "out": "y", "weight": 0.3}, {"to": "E", "weight": 0.2}]},
        {"id": "B", "trans": [{"to": "A", "out": "y", "weight": 0.6}, {"to": "B", "out": "x", "weight": 0.1}, {"to": "E", "weight": 0.3}]},
        {"id": "E"}]})
    # the same machine as a recogniser: its one tape is the input
    rec = Machine.fromJson(json.loads(json.dumps({"state": [{"id": st.name, "trans": [dict(to=t.dest, weight=t.weight, **({"in": t.out} if t.out else {})) for t in st.trans]} for st in gen.state]})))
    for mach, tape in ((gen, 1), (rec, 0)):
        em = EvaluatedMachine.fromMachine(mach, {})
        om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
        for ol in (0, 1, 23):
            seq = np.random.RandomState(ol).randint(1, 3, size=ol).astype(np.int32)
            x, y = (np.zeros(0, np.int32), seq) if tape else (seq, np.zeros(0, np.int32))
            V = dm.fill(capi.MB_VITERBI, x, y); F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
            assert capi.last_kernel_name().startswith("k_wide_")
            assert np.array_equal(V, om.viterbi(x, y))
            assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
            b = capi.DeviceBatch.from_pairs(dm, [(x, y)] * 2)
            assert close(b.forward(capi.MB_ROLLING), [F[-1, -1, -1]] * 2, 1e-9, 1e-12)


def test_boss_cli_hmmer_generator(capi, oracle_mod):
    """`boss --hmmer fn3.hmm --output-chars <protein> -L / -V` (target/boss.cpp:574-579): the profile is a one-tape machine
    (434 states) swept by the one-tape family; printed values against the oracle on the machine hmmer.py builds."""
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd.evalmachine import EvaluatedMachine
    hmm = golden_path("hmmer", "fn3.hmm")
    em = EvaluatedMachine.fromMachine(HmmerModel.fromFile(hmm).machine(True), {})
    om = oracle_mod.OracleMachine(em)
    prot = "PSAPTNLRVTDVTSTSVTLSWEPPPGPITGYRVEYREAGSEDWKEVTVPGSETSYTLTGLKPGTEYEVRVRAVNGAGEGPPSE"
    y = np.asarray(em.outputTokenizer.tokenize(list(prot)), np.int32); x = np.zeros(0, np.int32)
    got = json.loads(_boss(["--hmmer", hmm, "--output-chars", prot, "-L"]))
    assert capi.last_kernel_name().startswith("k_wide_")
    assert got[0][1] == prot and got[0][2] == float("%.6g" % om.loglike(x, y, oracle_mod.SUM_EXACT))
    got = json.loads(_boss(["--hmmer-global", hmm, "--output-chars", prot[:40], "-V"]))
    emg = EvaluatedMachine.fromMachine(HmmerModel.fromFile(hmm).machine(False), {})
    yg = np.asarray(emg.outputTokenizer.tokenize(list(prot[:40])), np.int32)
    assert got[0][2] == float("%.6g" % oracle_mod.OracleMachine(emg).viterbi(x, yg)[-1, -1, -1])


_COMM_ONE = r"""
import sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import numpy as np
from machineboss_amd import capi
counts = np.arange(7, dtype=np.float64) * 0.5
ll = C.c_double(-3.25)
assert capi.load().mb_allreduce_counts(None, counts.ctypes.data_as(C.POINTER(C.c_double)), counts.size, C.byref(ll)) == 0
comm = capi.Comm(capi.Comm.unique_id(), 1, 0)
try:
    got, gl = comm.allreduce_counts(counts.copy(), -3.25)
finally:
    comm.close()
assert np.array_equal(got, counts) and gl == -3.25
"""


def test_rccl_allreduce_entry_points(capi):
    """mb_comm_* / mb_allreduce_counts: the C-ABI route to the one collective of the path.  A one-rank communicator is all
    a single-GPU box can form -- the reduction is the identity there -- and a NULL communicator is a no-op; the N > 1
    arithmetic is covered by the gloo world-size-2 tests of the Python route (tests/test_distributed.py).  In a process of
    its own, as a C++ host would call it (and as test_rccl_allreduce_two_gpus does, one process per device): inside this
    pytest process -- two HIP runtimes once PyTorch has been imported, hundreds of modules loaded -- ncclCommInitRank of the
    image's RCCL aborted in one full-suite run out of two, taking every later test with it."""
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-c", _COMM_ONE, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]


_COMM_LONE = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from machineboss_amd import capi
capi.set_device(0)
uid = capi.Comm.unique_id()
t0 = time.time()
try:
    capi.Comm(uid, 2, 0)              # rank 1 never comes
    print("FORMED")
except Exception as e:
    print("ERROR %.1f %s" % (time.time() - t0, e))
sys.stdout.flush()
os._exit(0)                           # the abandoned bootstrap thread goes with the process
"""


@pytest.mark.gpu
def test_comm_init_gives_up_when_a_rank_never_arrives(capi):
    """First-contact hardening: mb_comm_init waits a BOUNDED time (MB_COMM_TIMEOUT_S) for the other ranks.
    One process asks for a two-rank communicator and nobody else ever calls in: the call must come back with an error that names
    the rank instead of hanging the job (shard.RankGroup then prints it and exits non-zero; nothing is retried in a process that has
    touched the GPU)."""
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MB_COMM_TIMEOUT_S="4")
    p = subprocess.run([sys.executable, "-c", _COMM_LONE, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith(("ERROR", "FORMED"))][-1]
    assert line.startswith("ERROR") and "rank 0 of 2" in line and "not formed within 4 s" in line, line
    assert 3.0 <= float(line.split()[1]) <= 60.0


_COMM_RANK = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
capi.set_device(rank)                                  # one process per GPU (mb_set_device before anything else)
if rank == 0:
    uid = capi.Comm.unique_id()                        # rank 0 makes the id and ships it by its own means (a file here)
    open(idfile + ".tmp", "wb").write(uid); os.rename(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile): break
        time.sleep(0.1)
    uid = open(idfile, "rb").read()
comm = capi.Comm(uid, world, rank)
em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(sys.argv[1], "tests", "golden", "preset", "protpsw.json")), None, useDefaults=True)
dm = capi.DeviceMachine(em)
pairs = [synth_tokens(5000 + k, 40 + k, 50, em.nInTok, em.nOutTok) for k in range(6)]
mine = pairs[rank::world]
counts, ll, _ = capi.DeviceBatch.from_pairs(dm, mine).counts()            # this rank's E-step (MachineCounts over its shard)
counts, ll = comm.allreduce_counts(counts, ll)                            # MachineCounts::operator+= over the ranks: ONE RCCL all-reduce
comm.close()
np.save(idfile + ".rank%d.npy" % rank, np.concatenate([counts, [ll]]))
"""


def test_rccl_allreduce_two_gpus(capi, machines, tmp_path):
    """The C-ABI collective on MORE than one GPU: two processes, one per device, bootstrap a communicator with
    mb_comm_unique_id / mb_comm_init and sum their E-step counts with mb_allreduce_counts (RCCL over xGMI).  Every rank must
    hold the single-process result (summation order differs: 1e-12).  Needs two visible GPUs: skipped on the
    one-GPU test box, runs wherever the driver's scaling tier has a node."""
    import subprocess, sys
    from conftest import ROOT
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs (the one-GPU box can only form a one-rank communicator: test_rccl_allreduce_entry_points)")
    idfile = str(tmp_path / "uid")
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, "-c", _COMM_RANK, ROOT, str(r), "2", idfile], env=env) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    m, em = machines("protpsw", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(5000 + k, 40 + k, 50, em.nInTok, em.nOutTok) for k in range(6)]
    ref_c, ref_ll, _ = capi.DeviceBatch.from_pairs(dm, pairs).counts()
    for r in range(2):
        got = np.load(idfile + ".rank%d.npy" % r)
        assert np.allclose(got[:-1], ref_c, rtol=1e-9, atol=1e-12) and abs(got[-1] - ref_ll) <= 1e-9 * abs(ref_ll)


@pytest.mark.parametrize("fp32", [0, 1])
def test_one_tape_family_long_sequences(capi, monkeypatch, fp32):
    """Lengths the CPU oracle would need minutes for: the one-tape family against the generic family (one barrier per
    silent level, exact fp64 log1p/exp) on an 8-node profile machine, 3 x 1500 nt -- Forward within 1e-6 relative for
    both arithmetic variants (fp64 columns; fp32 relative to the running column maximum), Viterbi scores and paths identical."""
    monkeypatch.setenv("MB_WIDE_FP32", str(fp32))
    m, em = _profile_machine(8)
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch(dm, *synth_batch(5, 3, 0, 1500, em.nInTok, em.nOutTok))
    res = {}
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
        capi.set_kernel(fam)
        try:
            res[fam] = (b.forward(capi.MB_ROLLING), b.forward(capi.MB_MATERIALISE), b.viterbi(paths=True))
            kern = capi.last_kernel_name()
        finally:
            capi.set_kernel(capi.KERNEL_AUTO)
        assert kern.startswith("k_generic") == (fam == capi.KERNEL_GENERIC)
    a, g = res[capi.KERNEL_AUTO], res[capi.KERNEL_GENERIC]
    assert close(a[0], g[0], 1e-6) and close(a[1], g[1], 1e-6) and close(a[0], a[1], FAST_REL, FAST_ABS)   # rolling: the three sequences are cut in two (k_onetape_join)
    monkeypatch.setenv("MB_ONETAPE_SPLIT", "0")
    assert close(b.forward(capi.MB_ROLLING), a[1], 1e-9, 1e-12)                                            # the plain rolling sweep is the materialised one without the stores
    assert np.array_equal(a[2][0], g[2][0]) and np.array_equal(a[2][1], g[2][1]) and np.array_equal(a[2][2], g[2][2])


def test_randomised_sweep(capi):
    """scripts/fuzz_gpu.py at a size for the suite: random two-tape and one-tape machines (duplicate edges, -inf weights,
    uneven silent levels, 1 to 900 states), ragged batches, automatically chosen family against the generic family and the
    oracle -- fills, rolling log-likelihoods, Viterbi paths, counts.  (250 cases were run by hand in round 1: 0 mismatches.)"""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_gpu.py"), "24", "4242"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "24 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_boss_cli_two_ranks_end_to_end(capi, tmp_path):
    """`torchrun -m machineboss_amd.boss` as the launcher sets it up (RANK / WORLD_SIZE / LOCAL_RANK in the environment):
    main() opens the process group itself, the pairs are sharded, rank 0 prints exactly what one process prints -- for
    --loglike (gathered in input order), --counts (all-reduced) and --train.  Two ranks share the one GPU of this box, so
    the exchange runs over gloo here (MB_DIST_BACKEND); on a multi-GPU node the same code path takes RCCL."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, "-m", "machineboss_amd.boss", golden_path("machine", "bitnoise.json"), "-D", golden_path("io", "seqpairlist.json")]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    for extra in (["-P", golden_path("io", "params.json"), "-L"], ["-P", golden_path("io", "params.json"), "-C"], ["-N", golden_path("io", "pqcons.json"), "-T"]):
        single = subprocess.run(base + extra, capture_output=True, text=True, cwd=ROOT)
        assert single.returncode == 0, single.stderr
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MB_DIST_BACKEND="gloo")
            procs.append(subprocess.Popen(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
        outs = [p.communicate(timeout=300) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        outs = [("".join(l for l in o.splitlines(True) if not l.startswith("[Gloo]")), e) for o, e in outs]   # gloo's own connection banner
        assert outs[1][0] == ""                                   # only rank 0 prints
        if "-T" in extra:
            a, b = json.loads(single.stdout), json.loads(outs[0][0])
            assert a.keys() == b.keys() and all(abs(a[k] - b[k]) <= 1e-6 * abs(a[k]) for k in a)
        else:
            assert outs[0][0] == single.stdout
        port += 1


def test_boss_cli_train_on_eight_ranks(capi, tmp_path):
    """`boss --train` sharded over EIGHT ranks (gloo, all on the one GPU of this box): the training set has fewer
    pairs than ranks -- some shards are empty --, every iteration's counts are summed over the ranks (MachineCounts::operator+=,
    src/counts.cpp:66-71), and rank 0 prints the parameters a single process fits."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, "-m", "machineboss_amd.boss", golden_path("machine", "bitnoise.json"), "-D", golden_path("io", "seqpairlist.json"), "-N", golden_path("io", "pqcons.json"), "-T"]
    single = subprocess.run(base, capture_output=True, text=True, cwd=ROOT)
    assert single.returncode == 0, single.stderr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MB_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    outs = ["".join(l for l in o.splitlines(True) if not l.startswith("[Gloo]")) for o, _ in outs]
    assert all(o == "" for o in outs[1:])                        # only rank 0 prints
    a, b = json.loads(single.stdout), json.loads(outs[0])
    assert a.keys() == b.keys() and all(abs(a[k] - b[k]) <= 1e-6 * abs(a[k]) for k in a)


@pytest.mark.parametrize("S,il,ol", [(300, 22, 39), (700, 27, 39), (257, 18, 59)])
def test_ahead_of_time_tile_kernel_many_states(capi, oracle_mod, monkeypatch, S, il, ol):
    """The interpreter kernel the tiled family falls back to without hiprtc (MB_MEDIUM_JIT=0), machines of more than 256
    states on short input sequences: the strip is narrowed to a workgroup of fewer than S / 4 threads, which the halo
    prefetch of that kernel did not cover (found by scripts/fuzz_gpu.py under MB_MEDIUM_JIT=0)."""
    from randmachine import random_machine, random_seq
    monkeypatch.setenv("MB_MEDIUM_JIT", "0")
    em = random_machine(S, 2, 3, S + il, density=1.6, silent_density=0.8)
    rng = np.random.RandomState(S)
    x, y = random_seq(rng, il, 2), random_seq(rng, ol, 3)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    V = dm.fill(capi.MB_VITERBI, x, y); kern = capi.last_kernel_name()
    F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
    assert kern.startswith("k_medium_tile")
    assert np.array_equal(V, om.viterbi(x, y))
    assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS) and close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y), (x[:5], y)])
    assert close(b.forward(capi.MB_ROLLING), [om.loglike(x, y, oracle_mod.SUM_EXACT), om.loglike(x[:5], y, oracle_mod.SUM_EXACT)], FAST_REL, FAST_ABS)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_dry_run(capi, scaling):
    """`python bench.py --gpus 2` as typed (the parent spawns the ranks before touching the GPU): the N > 1 path of the bench
    -- sharding, barrier + max-over-ranks timing, the EM leg with its all-reduce -- on a reduced workload.  Both ranks share
    the one GPU of this box and the collectives run over gloo (MB_BENCH_BACKEND / MB_BENCH_SHARE_DEVICE); on a multi-GPU
    node the same code takes RCCL."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MB_BENCH_BACKEND="gloo", MB_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "6", "--outlen", "700",
                        "--scaling", scaling], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    pairs_total = 12 if scaling == "weak" else 6
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2
    assert d["config"]["cells_per_gpu_per_step"] * 2 == pairs_total * 488 * 701 * 271
    assert abs(d["value"] - pairs_total * 488 * 701 * 271 / (d["ms_per_step"] * 1e-3) / 1e9) <= 1e-3 * d["value"]
    em = d["extra"]["em_iteration"]
    assert em["n_ranks_seen"] == 2 and abs(em["symbol_count_invariant"] - 1.0) < 1e-4
    # first-contact assertions of the N > 1 line, here over gloo on the one GPU: every rank seen once, the shards
    # add up to the batch, and the gathered log-likelihood checksum is the single-process one of the same synthetic pairs
    ck = d["extra"]["checks"]
    assert ck["ok"] and ck["n_ranks_seen"] == 2 and ck["ranks_distinct"] and ck["cells_all_ranks"] == ck["cells_expected"] == pairs_total * 488 * 701 * 271
    assert ck["pairs_all_ranks"] == ck["pairs_expected"] == pairs_total
    assert sorted(r["rank"] for r in d["extra"]["per_rank"]) == [0, 1] and sum(r["pairs"] for r in d["extra"]["per_rank"]) == pairs_total
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    emm = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "psw2dna.json")), None, useDefaults=True)
    single = capi.DeviceBatch(capi.DeviceMachine(emm), *synth_batch(4, pairs_total, 487, 700, emm.nInTok, emm.nOutTok)).forward(capi.MB_MATERIALISE)
    assert abs(ck["loglike_checksum_all_ranks"] - float(np.sum(single))) <= 1e-9 * abs(float(np.sum(single)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_eight_ranks_dry_run(capi, scaling, tmp_path):
    """N = 8 without a node: `python bench.py --gpus 8` as the driver's launcher would start it, the eight ranks sharing
    the one GPU of this box over gloo on a small workload.  What this exercises before a multi-GPU node ever sees the code: the shard
    arithmetic over 8 (weak: contiguous blocks; strong: longest-processing-time-first), the gathered checks (every rank seen once, cells
    and pairs add up, the log-likelihood checksum equals a single-process run), the EM leg's all-reduce over 8 ranks, eight processes
    compiling the same kernels COLD into one cache directory at once (written by rename), and pools sized for an eighth of the device
    (MB_MEM_FRACTION, set by shard.RankGroup for ranks that share a device)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MB_BENCH_BACKEND="gloo", MB_BENCH_SHARE_DEVICE="1", MB_JIT_CACHE_DIR=str(tmp_path / "jitcache"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MB_MEM_FRACTION"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--pairs", "4" if scaling == "weak" else "12", "--outlen", "1000",
                        "--scaling", scaling], capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    pairs_total = 32 if scaling == "weak" else 12
    assert d["n_gpus"] == 8 and d["scaling"] == scaling
    ck = d["extra"]["checks"]
    assert ck["ok"] and ck["n_ranks_seen"] == 8 and ck["ranks_distinct"] and ck["cells_all_ranks"] == ck["cells_expected"] == pairs_total * 488 * 1001 * 271
    assert ck["pairs_all_ranks"] == ck["pairs_expected"] == pairs_total
    per = d["extra"]["per_rank"]
    assert sorted(x["rank"] for x in per) == list(range(8)) and sum(x["pairs"] for x in per) == pairs_total
    if scaling == "strong": assert sorted(x["pairs"] for x in per) == [1, 1, 1, 1, 2, 2, 2, 2]      # 12 equal pairs dealt longest-first over 8
    else: assert all(x["pairs"] == 4 for x in per)
    em = d["extra"]["em_iteration"]
    assert em["n_ranks_seen"] == 8 and abs(em["symbol_count_invariant"] - 1.0) < 1e-4
    # eight cold compiles of the same sources into one directory: whole files only, no leftovers of the write-then-rename
    files = os.listdir(str(tmp_path / "jitcache"))
    assert files and all(f.endswith(".co") for f in files), files
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    emm = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "psw2dna.json")), None, useDefaults=True)
    single = capi.DeviceBatch(capi.DeviceMachine(emm), *synth_batch(4, pairs_total, 487, 1000, emm.nInTok, emm.nOutTok)).forward(capi.MB_MATERIALISE)
    assert abs(ck["loglike_checksum_all_ranks"] - float(np.sum(single))) <= 1e-9 * abs(float(np.sum(single)))


def _bench_json(args, env_extra, timeout=900):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 prints the one JSON line
    return json.loads(lines[0])


def test_bench_one_rank_rccl_communicator(capi):
    """The RCCL route of bench.py / boss.py (shard.RankGroup: unique id on rank 0 -> broadcast over gloo -> mb_comm_init ->
    mb_allreduce_counts on the library's own HIP runtime and stream, torch never on the GPU) with the one-rank communicator a
    one-GPU box can form: bootstrap, the EM leg's collective, teardown -- in a process of its own, as a host uses it."""
    d = _bench_json(["--steps", "1", "--warmup", "1", "--pairs", "4", "--outlen", "500", "--no-cpu", "--extra-em-only"], {"MB_BENCH_FORCE_COMM": "1"})
    em = d["extra"]["em_iteration"]
    assert d["n_gpus"] == 1 and em["n_ranks_seen"] == 1 and "RCCL through the C-ABI" in em["workload"]
    assert abs(em["symbol_count_invariant"] - 1.0) < 1e-4


@pytest.mark.parametrize("backend", ["rccl", "nccl"])
def test_bench_two_ranks_over_rccl(capi, backend):
    """`python bench.py --gpus 2` on TWO devices, the collectives over RCCL -- through the C-ABI on the library's runtime (the
    default) and through torch.distributed's NCCL backend.  Skips on the one-GPU test box; the first box with two devices runs
    it: both ranks seen, the per-rank cells sum to the batch, the all-reduced EM statistics keep the symbol-count invariant
    over BOTH ranks' pairs."""
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _bench_json(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "6", "--outlen", "700"], {"MB_BENCH_BACKEND": backend})
    assert d["n_gpus"] == 2 and d["config"]["cells_per_gpu_per_step"] * 2 == 12 * 488 * 701 * 271
    assert sum(r["cells_per_step"] for r in d["extra"]["per_rank"]) == 12 * 488 * 701 * 271
    em = d["extra"]["em_iteration"]
    assert em["n_ranks_seen"] == 2 and abs(em["symbol_count_invariant"] - 1.0) < 1e-4
    # first-contact assertions of the N > 1 line, here over gloo on the one GPU: every rank seen once, the shards
    # add up to the batch, and the gathered log-likelihood checksum is the single-process one of the same synthetic pairs
    ck = d["extra"]["checks"]
    assert ck["ok"] and ck["n_ranks_seen"] == 2 and ck["ranks_distinct"] and ck["cells_all_ranks"] == ck["cells_expected"] == pairs_total * 488 * 701 * 271
    assert ck["pairs_all_ranks"] == ck["pairs_expected"] == pairs_total
    assert sorted(r["rank"] for r in d["extra"]["per_rank"]) == [0, 1] and sum(r["pairs"] for r in d["extra"]["per_rank"]) == pairs_total
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    emm = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "psw2dna.json")), None, useDefaults=True)
    single = capi.DeviceBatch(capi.DeviceMachine(emm), *synth_batch(4, pairs_total, 487, 700, emm.nInTok, emm.nOutTok)).forward(capi.MB_MATERIALISE)
    assert abs(ck["loglike_checksum_all_ranks"] - float(np.sum(single))) <= 1e-9 * abs(float(np.sum(single)))


@pytest.mark.parametrize("family", ["small", "tiled", "onetape", "generic"])
def test_deterministic_counts_reproduce_bit_for_bit(capi, oracle_mod, machines, family):
    """MB_DETERMINISTIC=1: the reference's MachineCounts is a serial loop and reproduces bit for bit
    (src/counts.cpp:37-64); here posterior counts pass through LDS and global atomics whose order follows the scheduling.  In
    deterministic mode those accumulators are 64-bit fixed point (integer adds commute): repeated calls give IDENTICAL counts,
    which agree with the floating-point mode and with the oracle.  Every count path: small family, tiled family (flat program),
    one-tape family (lane = transition), generic fallback."""
    rng = np.random.RandomState(17)
    if family == "small":
        m, em = machines("protpsw", None, useDefaults=True, preset=True)
        pairs = [synth_tokens(50 + k, 90 + 7 * k, 110 - 5 * k, em.nInTok, em.nOutTok) for k in range(12)]
    elif family == "tiled":
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
        pairs = [synth_tokens(60 + k, 40 + 3 * k, 200 - 9 * k, em.nInTok, em.nOutTok) for k in range(10)]
    elif family == "onetape":
        m, em = _profile_machine(3)
        pairs = [(np.zeros(0, np.int32), rng.randint(1, 4, size=n).astype(np.int32)) for n in (150, 90, 300, 41)]
    else:
        m, em = machines("psw2dna", None, useDefaults=True, preset=True)
        pairs = [synth_tokens(70 + k, 30 + k, 60 + 2 * k, em.nInTok, em.nOutTok) for k in range(6)]
        capi.set_kernel(1)                                           # the generic family
    try:
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        plain = b.counts()[0].copy()
        capi.set_option("MB_DETERMINISTIC", "1")
        runs = [b.counts()[0].copy() for _ in range(4)]
    finally:
        capi.set_option("MB_DETERMINISTIC", None); capi.set_kernel(0)
    assert all(np.array_equal(r, runs[0]) for r in runs[1:])
    assert close(runs[0], plain, 1e-9, 1e-9)
    om = oracle_mod.OracleMachine(em)
    ref = np.zeros(em.nTransitions)
    for x, y in pairs:
        om.counts_add(x, y, ref, oracle_mod.SUM_EXACT)
    assert close(runs[0], ref, 1e-5, 1e-7)
    dm.close()


def test_counts_when_pytorch_is_imported_first(capi):
    """A host that imports PyTorch before this library makes libmbhip.so bind to the HIP runtime and hiprtc that PyTorch bundles
    (ROCm 7.0 here, /opt/rocm is 7.2): every kernel is then generated by another compiler.  Round 4 found the one place that
    mattered -- that compiler reports three AGPR-parked registers of the 482-state count kernel as spilled at every budget, the
    re-planning loop ran out of attempts and left the placement one step ahead of the code: counts of 1e19 -- in a full `pytest
    tests` run only (tests/test_distributed.py imports torch at collection).  In a process of its own: torch first, then the
    C4b count sweep against the oracle."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "torch_first_probe.py")], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, MB_JIT_CACHE="0"), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if "hiprtc |" in l][-1]
    errs = [float(tok.split()[-1]) for tok in line.split("|")[1:3]]
    assert max(errs) < 1e-5, line


def test_workspace_eviction_between_modes(capi, machines):
    """the memory budget counts cached workspaces as reclaimable, so a call that needs a big slot must be able to
    evict the pools earlier calls left behind.  Alternate Forward (pool 0), counts (pools 0 + 1) and Viterbi under an
    explicit budget that holds ONE call's buffers but not the sum of what the calls cache: every call still succeeds and
    gives the same results as with an unlimited budget."""
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)      # tiled family: fp64 matrices for every mode
    dm = capi.DeviceMachine(em)
    inTok, inOff, outTok, outOff = synth_batch(4, 6, 120, 700, em.nInTok, em.nOutTok)
    b = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)
    ref = (b.forward(capi.MB_MATERIALISE), b.viterbi(paths=False)[0], b.counts()[0])
    cells = b.cells() * 8
    capi.release_workspace()
    capi.set_memory_budget(int(2.2 * cells))      # counts need 2 matrices per pair; Forward + Viterbi + counts cached together would need 4
    try:
        for _ in range(2):
            assert np.array_equal(b.forward(capi.MB_MATERIALISE), ref[0])
            assert close(b.counts()[0], ref[2], 1e-9, 1e-12)
            assert np.array_equal(b.viterbi(paths=False)[0], ref[1])
    finally:
        capi.set_memory_budget(0)
        capi.release_workspace()

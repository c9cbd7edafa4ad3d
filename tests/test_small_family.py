"""The small-machine family (mb_small.cpp: lane = column, states in registers, tile-major matrices, traceback bytes,
count sweep over the Backward matrix) against the CPU oracle -- GPU tests -- and its host-side program / generated source
-- CPU tests.

Bars as in test_gpu_parity.py: Viterbi matrices, scores and tracebacks bit-exact; Forward / Backward within FAST_REL of
the oracle's exact-logsumexp mode (the correction term is evaluated in fp32) and within 1e-4 of the reference's table mode.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden_path, load_json
from machineboss_amd.seqgen import synth_tokens

FAST_REL = 2e-6
FAST_ABS = 2e-5


def close(a, b, rel, abs_=0.0):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    both_ninf = np.isneginf(a) & np.isneginf(b)
    fin = np.isfinite(a) & np.isfinite(b)
    if not np.all(both_ninf | fin):
        return False
    return bool(np.all(np.abs(a[fin] - b[fin]) <= abs_ + rel * np.abs(b[fin])))


@pytest.fixture(scope="module")
def capi():
    from machineboss_amd import capi as c
    if c.device_count() == 0:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return c


def _check_pairs(capi, oracle_mod, em, pairs, matrices=True):
    om = oracle_mod.OracleMachine(em)
    dm = capi.DeviceMachine(em)
    if matrices:
        for x, y in pairs:
            V = dm.fill(capi.MB_VITERBI, x, y)
            assert capi.last_kernel_name().startswith("k_small_")
            F = dm.fill(capi.MB_FORWARD, x, y); B = dm.fill(capi.MB_BACKWARD, x, y)
            assert np.array_equal(V, om.viterbi(x, y))
            assert close(F, om.forward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
            assert close(B, om.backward(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    ref = [om.loglike(x, y, oracle_mod.SUM_EXACT) for x, y in pairs]
    for flags in (capi.MB_MATERIALISE, capi.MB_ROLLING):
        ll = b.forward(flags)
        assert capi.last_kernel_name().startswith("k_small_")
        assert close(ll, ref, FAST_REL, FAST_ABS)
        assert close(ll, [om.loglike(x, y) for x, y in pairs], 1e-4, 1e-4)
    vll, off, edges = b.viterbi()
    for k, (x, y) in enumerate(pairs):
        V = om.viterbi(x, y)
        assert vll[k] == V[-1, -1, -1]
        got = edges[off[k]:off[k + 1]]
        if V[-1, -1, -1] > -math.inf:
            assert np.array_equal(got, om.traceback(x, y, V))
        else:
            assert len(got) == 0
    counts, s, cll = b.counts()
    ref_c = np.zeros(em.nTransitions); ref_s = 0.0
    for (x, y), l in zip(pairs, ref):
        ref_s += om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT) if l > -math.inf else l
    assert close(counts, ref_c, 1e-5, 1e-7)
    assert close(s, ref_s, FAST_REL, FAST_ABS) if math.isfinite(ref_s) else s == ref_s
    assert close(cll, ref, FAST_REL, FAST_ABS)


@pytest.mark.gpu
@pytest.mark.parametrize("name,shapes", [
    ("dnapsw", [(37, 53), (0, 0), (0, 9), (11, 0), (63, 1), (64, 64), (65, 130), (200, 70), (130, 260)]),
    ("protpsw", [(50, 50), (64, 3), (127, 129), (1, 200)]),
])
def test_small_presets(capi, oracle_mod, machines, name, shapes):
    """Single strip, several strips (inLen >= 64), several tiles (outLen + 64 > tile length), empty sequences."""
    m, em = machines(name, None, useDefaults=True, preset=True)
    pairs = [synth_tokens(300 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate(shapes)]
    _check_pairs(capi, oracle_mod, em, pairs)


@pytest.mark.gpu
@pytest.mark.parametrize("ts", ["64", "128"])
def test_small_tile_lengths(capi, oracle_mod, machines, monkeypatch, ts):
    """Every tile length cuts the sweeps differently (boundary records, halo rows): same results."""
    monkeypatch.setenv("MB_SMALL_TS", ts)
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    pairs = [synth_tokens(500 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate([(150, 333), (70, 64), (3, 190)])]
    _check_pairs(capi, oracle_mod, em, pairs, matrices=False)


@pytest.mark.gpu
@pytest.mark.parametrize("preset,jsub", [("dnapsw", None), ("dnapsw", "2"), ("dnapsw", "16"), ("protpsw", "64"), ("protpsw", None)])
def test_small_persistent_strips(capi, oracle_mod, machines, monkeypatch, preset, jsub):
    """PERSISTENT STRIPS (round 6; `src/api.cpp:31-66`, `target/boss.cpp:796-800`: one matrix object per pair): every
    strip of every pair is one wavefront of one grid and sweeps its whole strip, the halo column between strips is its own hand-over
    (a sentinel the host writes, agent-scope stores and loads; mb_small.cpp HALO_EMPTY).  The library's own choice for batches of at
    most 1 024 strips.  Against the launch-by-launch sweep (MB_SMALL_ONE_LAUNCH=0), bit for bit: log-likelihoods, Forward / Backward /
    Viterbi matrices, scores and paths, counts (Backward fill through strips, count sweep launch by launch); against the oracle on the
    short pairs; a batch too large for the rule runs launch by launch; hand-over block lengths 2 ... 64 steps (MB_SMALL_JSUB)."""
    m, em = machines(preset, None, useDefaults=True, preset=True)
    L = 1000 if preset == "dnapsw" else 400
    big = [synth_tokens(900, L, L, em.nInTok, em.nOutTok)]
    ragged = [synth_tokens(910 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate([(150, 333), (70, 64), (3, 190), (0, 5), (300, 129), (64, 0), (63, 1), (65, 200)])]
    if jsub: monkeypatch.setenv("MB_SMALL_JSUB", jsub)
    out = {}
    for one in ("default", "0"):
        if one == "0": monkeypatch.setenv("MB_SMALL_ONE_LAUNCH", "0")
        else: monkeypatch.delenv("MB_SMALL_ONE_LAUNCH", raising=False)
        dm = capi.DeviceMachine(em)
        r = {}
        for name, pairs in (("big", big), ("ragged", ragged)):
            b = capi.DeviceBatch.from_pairs(dm, pairs)
            r[name + ".roll"] = b.forward(capi.MB_ROLLING); r[name + ".roll.launches"] = capi.last_launch_count()
            r[name + ".mat"] = b.forward(capi.MB_MATERIALISE)
            r[name + ".vit"] = b.viterbi(); r[name + ".vit.launches"] = capi.last_launch_count()
            r[name + ".cnt"] = b.counts()
        x, y = ragged[0]
        for mode in (capi.MB_FORWARD, capi.MB_BACKWARD, capi.MB_VITERBI): r["fill%d" % mode] = dm.fill(mode, x, y)
        out[one] = r
        dm.close()
    a, b_ = out["default"], out["0"]
    assert a["big.roll.launches"] == 1 and a["ragged.roll.launches"] == 1 and b_["big.roll.launches"] > 10, (a["big.roll.launches"], b_["big.roll.launches"])
    for k in a:
        if k.endswith(".launches"): continue
        if k.endswith(".vit"):
            for u, v in zip(a[k], b_[k]): assert np.array_equal(np.asarray(u), np.asarray(v)), k
        elif k.endswith(".cnt"): assert close(a[k][0], b_[k][0], 1e-9, 1e-12) and close(a[k][1], b_[k][1], 1e-12)
        else: assert np.array_equal(np.asarray(a[k]), np.asarray(b_[k])), k
    om = oracle_mod.OracleMachine(em)
    for k, (x, y) in enumerate(ragged):
        assert close(a["ragged.roll"][k], om.loglike(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        V = om.viterbi(x, y); vll, off, edges = a["ragged.vit"]
        assert vll[k] == V[-1, -1, -1]
        if V[-1, -1, -1] > -math.inf: assert np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, V))
    # more strips than the rule admits: launch by launch, by the library's own choice
    if jsub is None and preset == "dnapsw":
        monkeypatch.delenv("MB_SMALL_ONE_LAUNCH", raising=False)
        dm = capi.DeviceMachine(em)
        many = [synth_tokens(3000 + k, 300, 200, em.nInTok, em.nOutTok) for k in range(260)]      # 260 x 5 strips
        bm = capi.DeviceBatch.from_pairs(dm, many)
        llm = bm.forward(capi.MB_ROLLING)
        assert capi.last_launch_count() > 1
        monkeypatch.setenv("MB_SMALL_ONE_LAUNCH", "2")
        dm2 = capi.DeviceMachine(em)
        assert np.array_equal(capi.DeviceBatch.from_pairs(dm2, many).forward(capi.MB_ROLLING), llm) and capi.last_launch_count() == 1
        dm2.close(); dm.close()


@pytest.mark.gpu
def test_small_persistent_strips_fall_back_when_a_row_never_arrives(tmp_path):
    """A strip that waits longer than its bound for a halo row (a shared device; here: a bound of one microsecond) raises the sweep's
    error word; the host latches the one-grid forms off for the process and runs the sweep again launch by launch -- the caller gets the
    right answer and a warning, and the sweeps that follow no longer try.  (A process of its own: the latch is per process.)"""
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
em = EvaluatedMachine.fromMachine(Machine.fromFile(%r), None, useDefaults=True)
pairs = [synth_tokens(900, 600, 700, em.nInTok, em.nOutTok)]
dm = capi.DeviceMachine(em); b = capi.DeviceBatch.from_pairs(dm, pairs)
capi.set_option("MB_SMALL_ONE_LAUNCH", "0"); ref = b.forward(capi.MB_ROLLING); vref = b.viterbi(); n0 = capi.last_launch_count()
capi.set_option("MB_SMALL_ONE_LAUNCH", "2"); ok = b.forward(capi.MB_ROLLING); n1 = capi.last_launch_count()
capi.set_option("MB_SMALL_ONE_LAUNCH_TIMEOUT_US", "1")
got = b.forward(capi.MB_ROLLING); n2 = capi.last_launch_count()
capi.set_option("MB_SMALL_ONE_LAUNCH_TIMEOUT_US", "0")
again = b.forward(capi.MB_ROLLING); n3 = capi.last_launch_count(); vit = b.viterbi()
assert n0 > 10 and n1 == 1 and n2 > 10 and n3 > 10, (n0, n1, n2, n3)
assert np.array_equal(ref, ok) and np.array_equal(ref, got) and np.array_equal(ref, again)
assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(vref, vit))
print("FELL BACK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), golden_path("preset", "dnapsw.json"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FELL BACK" in r.stdout, (r.stdout + r.stderr)[-2000:]
    assert "run again launch by launch" in r.stderr


@pytest.mark.gpu
def test_small_one_launch_sweeps(capi, oracle_mod, machines, monkeypatch):
    """ONE launch for a whole sweep (round 6; `src/api.cpp:31-66`, `target/boss.cpp:796-800`: one matrix object per
    pair): a single 1 kb x 1 kb dnapsw pair was a chain of 47 dependent launches -- now every tile of the sweep is in one grid, in
    wavefront order, and waits for the tiles it reads from through a "done" word per tile.  Same results as launch by launch
    (MB_SMALL_ONE_LAUNCH=0), bit for bit: log-likelihoods, Forward / Backward / Viterbi matrices, Viterbi scores and paths; the oracle
    on the short pairs; envelopes (tiles that do not run have no "done" word to wait for); one launch counted."""
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    big = [synth_tokens(900, 1000, 1000, em.nInTok, em.nOutTok)]
    ragged = [synth_tokens(910 + k, a, b, em.nInTok, em.nOutTok) for k, (a, b) in enumerate([(150, 333), (70, 64), (3, 190), (0, 5), (300, 129)])]
    out = {}
    for one in ("1", "0"):
        monkeypatch.setenv("MB_SMALL_ONE_LAUNCH", one)
        dm = capi.DeviceMachine(em)
        r = {}
        for name, pairs in (("big", big), ("ragged", ragged)):
            b = capi.DeviceBatch.from_pairs(dm, pairs)
            r[name + ".roll"] = b.forward(capi.MB_ROLLING); r[name + ".roll.launches"] = capi.last_launch_count()
            r[name + ".mat"] = b.forward(capi.MB_MATERIALISE)
            r[name + ".vit"] = b.viterbi(); r[name + ".vit.launches"] = capi.last_launch_count()
            r[name + ".cnt"] = b.counts()
        x, y = ragged[0]
        for mode in (capi.MB_FORWARD, capi.MB_BACKWARD, capi.MB_VITERBI): r["fill%d" % mode] = dm.fill(mode, x, y)
        out[one] = r
        dm.close()
    a, b_ = out["1"], out["0"]
    assert a["big.roll.launches"] == 1 and b_["big.roll.launches"] > 20, (a["big.roll.launches"], b_["big.roll.launches"])
    for k in a:
        if k.endswith(".launches"): continue
        if k.endswith(".vit"):
            for u, v in zip(a[k], b_[k]): assert np.array_equal(np.asarray(u), np.asarray(v)), k
        elif k.endswith(".cnt"): assert close(a[k][0], b_[k][0], 1e-9, 1e-12) and close(a[k][1], b_[k][1], 1e-12)      # (the count sweep itself runs launch by launch; its Backward fill in one)
        else: assert np.array_equal(np.asarray(a[k]), np.asarray(b_[k])), k
    om = oracle_mod.OracleMachine(em)
    for k, (x, y) in enumerate(ragged):
        assert close(a["ragged.roll"][k], om.loglike(x, y, oracle_mod.SUM_EXACT), FAST_REL, FAST_ABS)
        V = om.viterbi(x, y); vll, off, edges = a["ragged.vit"]
        assert vll[k] == V[-1, -1, -1] and np.array_equal(edges[off[k]:off[k + 1]], om.traceback(x, y, V))
    # a path envelope (quirk Q1: what an aligned pair gets): dead tiles are not in the grid
    monkeypatch.setenv("MB_SMALL_ONE_LAUNCH", "1")
    x, y = synth_tokens(950, 200, 260, em.nInTok, em.nOutTok)
    rng = np.random.RandomState(5)
    cols = []; i = o = 0
    while i < len(x) or o < len(y):
        step = rng.randint(3) if (i < len(x) and o < len(y)) else (1 if i < len(x) else 2)
        if step == 0: cols.append((i, o)); i += 1; o += 1
        elif step == 1: cols.append((i, None)); i += 1
        else: cols.append((None, o)); o += 1
    envStart = np.zeros(len(y) + 1, np.int32); envEnd = np.zeros(len(y) + 1, np.int32)
    # the band of width 8 around the alignment's path
    ii = oo = 0; lo = [len(x) + 1] * (len(y) + 1); hi = [0] * (len(y) + 1)
    def mark(ii, oo):
        lo[oo] = min(lo[oo], ii); hi[oo] = max(hi[oo], ii)
    mark(0, 0)
    for ci, co in cols:
        if ci is not None: ii += 1
        if co is not None: oo += 1
        mark(ii, oo)
    for oo in range(len(y) + 1): envStart[oo] = max(0, lo[oo] - 8); envEnd[oo] = min(len(x) + 1, hi[oo] + 9)
    res = {}
    for one in ("1", "0"):
        monkeypatch.setenv("MB_SMALL_ONE_LAUNCH", one)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        b.set_envelopes([(envStart, envEnd)])
        res[one] = (b.forward(capi.MB_ROLLING), b.viterbi())
        dm.close()
    assert np.array_equal(res["1"][0], res["0"][0]) and all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(res["1"][1], res["0"][1]))
    ref = om.loglike_env(x, y, envStart, envEnd, oracle_mod.SUM_EXACT) if hasattr(om, "loglike_env") else None
    if ref is not None: assert close(res["1"][0][0], ref, FAST_REL, FAST_ABS)


@pytest.mark.gpu
@pytest.mark.parametrize("S,seed", [(2, 1), (3, 2), (5, 3), (8, 4), (12, 5), (16, 6), (7, 7), (9, 8)])
def test_small_random_machines(capi, oracle_mod, S, seed):
    """Random machines: odd state counts (8-byte chunks), duplicate edges, several edges per label, match edges, -inf weights."""
    from randmachine import random_machine, random_seq
    em = random_machine(S, 3, 4, seed, allow_inf=(seed % 3 == 0))
    rng = np.random.RandomState(seed)
    pairs = [(random_seq(rng, il, 3), random_seq(rng, ol, 4)) for il, ol in [(25, 31), (0, 9), (80, 45), (5, 140)]]
    _check_pairs(capi, oracle_mod, em, pairs)


@pytest.mark.gpu
def test_small_reference_goldens(capi, oracle_mod, machines):
    """The reference's tiny machines (bitnoise: 1 state... bitstutter-noise) run through this family too when they qualify."""
    p = load_json("io", "params.json")
    for name, il, ol in (("bitnoise", 3, 3), ("bitstutter-noise", 5, 9)):
        m, em = machines(name, p)
        x, y = synth_tokens(7, il, ol, em.nInTok, em.nOutTok)
        _check_pairs(capi, oracle_mod, em, [(x, y)])


@pytest.mark.gpu
def test_small_vs_medium_large(capi, machines):
    """Size-independent check at a larger shape: this family and the tiled family give the same Viterbi scores and paths
    (bit for bit), Forward log-likelihoods within the fp32-correction tolerance, counts within 1e-6."""
    m, em = machines("dnapsw", None, useDefaults=True, preset=True)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(900 + k, 700 + 13 * k, 900 - 7 * k, em.nInTok, em.nOutTok) for k in range(6)]
    b = capi.DeviceBatch.from_pairs(dm, pairs)
    res = {}
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_MEDIUM):
        capi.set_kernel(fam)
        try:
            res[fam] = (b.forward(capi.MB_ROLLING), b.forward(capi.MB_MATERIALISE), b.viterbi(), b.counts(), capi.last_kernel_name())
        finally:
            capi.set_kernel(capi.KERNEL_AUTO)
    a, g = res[capi.KERNEL_AUTO], res[capi.KERNEL_MEDIUM]
    assert a[4].startswith("k_small_") and "k_medium" in g[4]
    assert close(a[0], g[0], FAST_REL) and close(a[1], g[1], FAST_REL) and close(a[0], a[1], 1e-12)
    assert np.array_equal(a[2][0], g[2][0]) and np.array_equal(a[2][1], g[2][1]) and np.array_equal(a[2][2], g[2][2])
    assert close(a[3][0], g[3][0], 1e-5, 1e-6) and close(a[3][1], g[3][1], FAST_REL)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5013, 5229, 5272, 5001, 5005, 5009])
def test_small_envelope_batches_regressions(capi, oracle_mod, seed):
    """Cases of scripts/fuzz_env_gpu.py (random machine, ragged batch, some pairs under the path-area envelope of a random
    alignment), among them the three that exposed bugs in round 2: Backward and Forward programs of an asymmetric machine
    keep different numbers of values per halo row / boundary record and share the buffers of a count call (5013, 5272);
    a 12-state machine whose count sweep does not fit the registers of 3 wavefronts per SIMD (5229)."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_env_gpu", os.path.join(ROOT, "scripts", "fuzz_env_gpu.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.run_case(seed) == {}


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,kernel", [("dnapsw", 100, 0), ("dnapsw", 128, 0), ("dnapsw", 129, 0), ("dnapsw", 191, 0), ("protpsw", 257, 0),
                                           ("dnapsw", 129, 3), ("dnapsw", 129, 1)])
def test_gapless_path_envelope_across_strip_and_block_boundaries(capi, oracle_mod, machines, monkeypatch, name, n, kernel):
    """Round-2 ADVICE (high): a gapless alignment of >= 64 columns under its own path envelope (width 0 -- what the reference's
    Envelope(seqPair) gives an aligned pair, quirk Q1) crosses every strip boundary with a MATCH at (64a, 64a), i.e. at lane 0
    of strip a at the first step of block b = a (64-step tiles), while block (a, a - 1) holds no cell of the envelope and is
    not launched: lane 0's diagonal predecessor (64a - 1, 64a - 1) then has to come from the halo column of strip a - 1, not
    from the boundary record the skipped block never wrote.  Before the fix every sweep returned -inf here.  Also a wider
    band whose lower edge is that same diagonal, the Backward frame (mirrored strips), the tiled and the generic family."""
    from machineboss_amd.seqpair import Envelope
    monkeypatch.setenv("MB_SMALL_TS", "64")
    m, em = machines(name, None, useDefaults=True, preset=True)
    om = oracle_mod.OracleMachine(em); dm = capi.DeviceMachine(em)
    x, y = synth_tokens(70 + n, n, n, em.nInTok, em.nOutTok)
    cols = [("a", "b")] * n
    capi.set_kernel(kernel)
    try:
        for env in (Envelope.pathAreaEnvelope(cols, 0), Envelope.pathEnvelope(cols), Envelope.pathAreaEnvelope(cols[:n // 2] + [("a", "")] * 3 + [("", "b")] * 3 + cols[n // 2 + 3:], 0)):
            with oracle_mod.envelope(env.inStart, env.inEnd):
                Fo = om.forward(x, y, oracle_mod.SUM_EXACT); Bo = om.backward(x, y, oracle_mod.SUM_EXACT); Vo = om.viterbi(x, y)
                po = om.traceback(x, y, Vo)
                ref_c = np.zeros(em.nTransitions); ref_ll = om.counts_add(x, y, ref_c, oracle_mod.SUM_EXACT)
            assert math.isfinite(Fo[-1, -1, -1]) and math.isfinite(Vo[-1, -1, -1])
            F = dm.fill(capi.MB_FORWARD, x, y, 0, env.inStart, env.inEnd)
            if kernel == 0:
                assert capi.last_kernel_name().startswith("k_small_")
            B = dm.fill(capi.MB_BACKWARD, x, y, 0, env.inStart, env.inEnd); V = dm.fill(capi.MB_VITERBI, x, y, 0, env.inStart, env.inEnd)
            assert np.array_equal(V, Vo)
            assert close(F, Fo, FAST_REL, FAST_ABS) and close(B, Bo, FAST_REL, FAST_ABS)
            b = capi.DeviceBatch.from_pairs(dm, [(x, y), (x[:70], y[:70])])
            b.set_envelopes([(env.inStart, env.inEnd), None])
            for flags in (capi.MB_MATERIALISE, capi.MB_ROLLING):
                assert close(b.forward(flags)[0], Fo[-1, -1, -1], FAST_REL, FAST_ABS)
            vll, off, edges = b.viterbi()
            assert vll[0] == Vo[-1, -1, -1] and np.array_equal(edges[off[0]:off[1]], po)
            b1 = capi.DeviceBatch.from_pairs(dm, [(x, y)])
            b1.set_envelopes([(env.inStart, env.inEnd)])
            counts, s_ll, cll = b1.counts()
            assert close(counts, ref_c, 1e-5, 1e-7) and close(cll[0], ref_ll, FAST_REL, FAST_ABS)
    finally:
        capi.set_kernel(0)


# ---- CPU: program structure and generated source ---------------------------------------------------------------------------
def test_small_source_compiles_for_gfx950(tmp_path, machines):
    """Every mode of the generated kernel cross-compiles for gfx950 without spills (hipcc needs no GPU)."""
    from machineboss_amd import capi
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    m, em = machines("protpsw", None, useDefaults=True, preset=True)
    for mode, backward, mat in ((0, False, True), (0, True, True), (0, False, False), (1, False, True), (2, False, False), (3, False, False)):
        src = str(tmp_path / ("k%d%d%d.hip" % (mode, backward, mat)))
        capi.debug_small_source(em, src, mode=mode, backward=backward, materialise=mat)
        full = src.replace(".hip", "_full.hip")
        open(full, "w").write("#include <hip/hip_runtime.h>\n" + open(src).read())
        asm = src.replace(".hip", ".s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics",
                               "--cuda-device-only", "-S", "-o", asm, full], stderr=subprocess.DEVNULL)
        txt = open(asm).read()
        assert ".vgpr_spill_count: 0" in txt and "wave_shr:1" in txt
        assert not [l for l in txt.splitlines() if l.startswith("\tflat_")]      # LDS / global accesses stay typed (no flat fallback)


def test_small_count_kernel_lds_block_holds_its_tables(tmp_path):
    """The per-wavefront LDS block of the count sweep (tokens, envelope rows, halo rows, lane-private usage rows, the
    output-only usage sums) as the generated source lays it out fits the size the host reserves for it (JWAVEDBL): it was
    one float short when (output-only tables) x (alphabet + 1) is even, and the last usage sum landed on the next
    wavefront's first token (scripts/fuzz_env_gpu.py seed 5229)."""
    import re
    from machineboss_amd import capi
    from randmachine import random_machine
    checked = 0
    for S, nIn, nOut, seed in ((12, 1, 1, 5229), (8, 2, 3, 11), (5, 3, 1, 12), (3, 1, 2, 13), (16, 2, 2, 14), (7, 1, 3, 15)):
        em = random_machine(S, nIn, nOut, seed, density=2.0, silent_density=1.0)
        src = str(tmp_path / ("c%d.hip" % seed))
        try:
            capi.debug_small_source(em, src, mode=3, materialise=False)
        except capi.MbError:
            continue
        d = {k: int(v) for k, v in re.findall(r"#define (J[A-Z]+) (-?\d+)", open(src).read())}
        floats = 64 * d["JROWF"] + (d["JROWF"] & 1) + d["JOUTACC"]      # rowL rows, then outAcc (started (JROWF & 1) floats further on)
        assert 32 + 64 + 64 * d["JHP"] + (floats + 1) // 2 <= d["JWAVEDBL"], (S, nIn, nOut, d)
        checked += 1
    assert checked >= 4


def test_small_rejects_what_it_cannot_run(machines):
    """Machines outside the family's envelope (too many states, one-tape) are refused by the generator, so the library
    falls back to the other families."""
    from machineboss_amd import capi
    m, em = machines("psw2dna", None, useDefaults=True, preset=True)
    with pytest.raises(capi.MbError, match="does not qualify"):
        capi.debug_small_source(em, "/tmp/never.hip")

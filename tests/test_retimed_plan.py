"""The retimed sweep of the one-tape family (machineboss_amd/csrc/mb_wide.hip: wide_ret_build -> k_wide_retimed), checked
WITHOUT a device: `mb_debug_wide_retimed` hands back the record streams exactly as the kernel reads them, `simulate`
below restates what the kernel does with them -- a period of slots, every lane folding `(ring[src] + w) + penalty`, lane
groups reduced at a round's end, the node stored into the ring vector of its own column -- and the cells it produces are
compared with the oracle (src/viterbi.cpp:18-43 for max: bit for bit; src/forward.defs.h:23-49 for sum).  What this pins: the
time offsets (a value is never read before it is written nor after its ring vector was reused), the relay entries, the
penalty table (tokens, the seed), the rotation streams and the end-of-round words.  The arithmetic of the kernel itself is
covered on the GPU (tests/test_gpu_parity.py::test_one_tape_retimed_sweep)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import golden_path
from randmachine import random_machine

NO_DST = 0x3ffff


def simulate(prog, seq, backward, mode_max, tb=False, part=None):
    """cells[column][state] of one sequence from the record streams (the kernel's data flow, slot by slot).
    part = (X, cells, codes): `prog` is one PART of a machine cut for k workgroups per sequence (capi.debug_wide_parts): its imports
    come from the exchange array X[column][exchange column] through the penalty table's tail (the kernel's wait for a value is the
    assertion that an earlier part has written it), its exports go there, its own states' cells go to their machine columns.
    tb (forward max program built for traceback codes): also codes[column][state] -- the place of the cell's first maximal
    candidate in its node's list, as k_wide_retimed<1,.,codes> keeps it: a lane remembers the slot of its first strictly greater
    candidate within the round, the place is slot * group + lane within the group, and among the lanes that hold the group's
    maximum the smallest place wins."""
    W, NB, NVs, kMax, rowLen, inL2 = (prog[k] for k in ("lanes", "NB", "NVs", "kMax", "rowLen", "inL2"))
    S = prog["Sloc"] if part else prog["S"]
    L = len(seq)
    V = np.full(NB * NVs, -np.inf)
    for b in range(NB):
        V[b * NVs + S + 1] = 0.0
    if part:
        X, cells, codes = part
        gmap = prog["gmap"]
        assert prog["nPen"] == (kMax + 1) * rowLen and prog["nImp"] <= W and prog["expBase"] == S + prog["nImp"] + 2
    else:
        cells = np.full((L + 1, S), np.nan)
        codes = np.full((L + 1, S), -1, np.int64)
        gmap = np.arange(S)
    best = np.zeros(W, np.int64)
    tok_at = lambda c: (int(seq[L - c]) if backward else int(seq[c - 1])) if 1 <= c <= L else 0
    lanes = np.arange(W)
    for t in range(L + 1 + kMax):
        cm = t % NB
        pen = np.full((kMax + 1, rowLen), -np.inf)
        for kt in range(kMax + 1):
            c = t - kt
            pen[kt, 0] = 0.0
            if c == 0: pen[kt, rowLen - 1] = 0.0
            if c >= 1 and 0 < tok_at(c) < rowLen - 1: pen[kt, tok_at(c)] = 0.0
        pen = pen.reshape(-1)
        if part:      # the imports of the period's newest column, behind the (ktau, token) entries
            imp = X[t, prog["impIdx"]] if t <= L else np.full(prog["nImp"], -np.inf)
            assert not np.any(np.isnan(imp)), "an import is read before an earlier part wrote it"
            pen = np.concatenate([pen, imp])
        m = np.full(W, -np.inf); ssum = np.zeros(W)
        slot_in_round = 0
        for rec in prog["records"][cm]:
            src = rec["src"].astype(np.int64)
            entry = (src >> 13) if inL2 else ((src >> 14) >> 3)
            assert inL2 or np.all(((src >> 14) & 7) == 0)
            cand = (V[entry] + rec["w"]) + pen[src & 0x1fff]
            if part: cand = cand + rec["w2"]                      # a two-transition candidate's second rounded add (+ 0.0 otherwise)
            if tb: best = np.where(cand > m, slot_in_round, best); slot_in_round += 1      # strict >: the first maximum
            if mode_max: m = np.maximum(m, cand)
            else:      # max and sum of exp relative to it, as the kernel keeps them (in fp64 here)
                new = np.maximum(m, cand)
                with np.errstate(invalid="ignore"):
                    ssum = np.where(np.isneginf(new), 0.0, ssum * np.exp(np.where(np.isneginf(m), -np.inf, m - new)) + np.exp(np.where(np.isneginf(cand), -np.inf, cand - new)))
                m = new
            flags = int(rec["pad"][0])
            if not flags & 0x80000000: assert not np.any(rec["pad"]); continue
            pad = rec["pad"].astype(np.int64)
            assert np.all(pad & 0x80000000) and np.all(((pad >> 30) & 1) == ((flags >> 30) & 1))
            log2g = (pad >> 26) & 7
            heads = lanes[(pad & NO_DST) != NO_DST]
            # a wavefront's first lane tells whether its groups have different sizes (masked reduction in the kernel)
            for w0 in range(0, W, 64):
                hw = heads[(heads >= w0) & (heads < w0 + 64)]
                mixed = bool(hw.size) and bool(np.any(log2g[hw] != log2g[w0]))
                assert mixed == bool((int(pad[w0]) >> 29) & 1)
            for l in heads:
                g = 1 << int(log2g[l])
                assert l % g == 0 and np.all(log2g[l:l + g] == log2g[l])
                if mode_max: res = float(np.max(m[l:l + g]))
                if tb:
                    place = (best[l:l + g] << int(log2g[l])) | (lanes[l:l + g] & (g - 1))
                    code = int(np.min(place[m[l:l + g] == res]))
                if not mode_max:
                    mx = float(np.max(m[l:l + g]))
                    res = -math.inf if mx == -math.inf else mx + math.log(float(np.sum(ssum[l:l + g] * np.exp(np.where(np.isneginf(m[l:l + g]), -np.inf, m[l:l + g] - mx)))))
                x, kq, vec = int(pad[l] & NO_DST), int((pad[l] >> 20) & 63), int((pad[l] >> 18) & 3)
                c = t - kq if backward else t - kMax + kq
                if 0 <= c <= L:
                    assert vec == c % NB                          # the node's own column's vector
                    V[vec * NVs + x] = res
                    if part and prog["expBase"] <= x < prog["expBase"] + prog["nExp"]:      # (relay entries follow the exports)
                        e = prog["expIdx0"] + x - prog["expBase"]
                        assert np.isnan(X[c, e])
                        X[c, e] = res
                    if x < S:
                        assert np.isnan(cells[L - c if backward else c, gmap[x]])      # every cell exactly once
                        cells[L - c if backward else c, gmap[x]] = res
                        if tb: assert 0 <= code < 256; codes[c, gmap[x]] = code
            m[:] = -np.inf; ssum[:] = 0.0
            best[:] = 0; slot_in_round = 0
    if part: return None
    assert not np.any(np.isnan(cells))
    return (cells, codes) if tb else cells


def simulate_parts(pp, seq, backward, mode_max, tb=False):
    """The k parts of one machine, one after the other in the order their workgroups are numbered (a part reads what LOWER parts
    export): cells (and codes) of the whole machine."""
    L = len(seq)
    X = np.full((L + 1, max(pp["nExp"], 1)), np.nan)
    cells = np.full((L + 1, pp["S"]), np.nan)
    codes = np.full((L + 1, pp["S"]), -1, np.int64)
    owners = np.concatenate([p["gmap"] for p in pp["parts"]])
    assert np.array_equal(np.sort(owners), np.arange(pp["S"]))            # every state belongs to exactly one part
    assert sum(p["nExp"] for p in pp["parts"]) == pp["nExp"] and sum(p["resultEntry"] >= 0 for p in pp["parts"]) == 1
    for p in pp["parts"]:
        assert np.all(p["impIdx"] < p["expIdx0"])                         # imports come from lower parts only
        simulate(p, seq, backward, mode_max, tb, part=(X, cells, codes))
    assert not np.any(np.isnan(cells)) and not np.any(np.isnan(X[:, :pp["nExp"]]))
    res = [p for p in pp["parts"] if p["resultEntry"] >= 0][0]
    assert res["gmap"][res["resultEntry"]] == (0 if backward else pp["S"] - 1)
    return (cells, codes) if tb else cells


def walk_codes(prog, codes, L):
    """k_onetape_traceback_codes restated: from (L, end state) back to (0, start state) through the decode tables; the path as
    global edge ids, start -> end."""
    S = prog["S"]; off, ent, eid = prog["tbOff"], prog["tbEntry"], prog["inEid"]
    c, s, path = L, S - 1, []
    while not (c == 0 and s == 0):
        code = int(codes[c, s]); o0, o1 = int(off[s]), int(off[s + 1])
        assert 0 <= code < o1 - o0, "dead end"
        e = int(ent[o0 + code])
        if e == 0xFFFFFFFF:
            assert c == 0
            break
        path.append(int(eid[e >> 16]))
        s = e & 0x7fff
        if e & 0x8000:
            assert c > 0
            c -= 1
        assert len(path) <= (L + 1) * (S + 1)
    return np.asarray(path[::-1], np.int64)


def _machines():
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd import algebra as A
    gen = Machine.fromJson({"state": [
        {"id": "S", "trans": [{"to": "A"}, {"to": "B", "weight": 0.25}]},
        {"id": "A", "trans": [{"to": "A", "out": "x", "weight": 0.5}, {"to": "B", "out": "y", "weight": 0.3}, {"to": "E", "weight": 0.2}]},
        {"id": "B", "trans": [{"to": "A", "out": "y", "weight": 0.6}, {"to": "B", "out": "x", "weight": 0.1}, {"to": "E", "weight": 0.3}]},
        {"id": "E"}]})
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm"))
    comp = A.composeLeftToRight([h.truncated(2).machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    return {"tiny": EvaluatedMachine.fromMachine(gen, {}),
            "fn3-10": EvaluatedMachine.fromMachine(h.truncated(10).machine(True), {}),
            "composite-2": EvaluatedMachine.fromMachine(comp, None, useDefaults=True),
            "random-40": random_machine(40, 0, 3, 7, density=2.0, silent_density=1.5, allow_inf=True),
            "random-recogniser-25": random_machine(25, 2, 0, 11, density=1.5, silent_density=1.0, allow_inf=False)}


@pytest.mark.parametrize("name", ["tiny", "fn3-10", "composite-2", "random-40", "random-recogniser-25"])
@pytest.mark.parametrize("knobs", [{}, {"MB_WIDE_GLOBAL_VECTORS": "1"}, {"MB_WIDE_RETIMED_PERIOD": "+2", "MB_WIDE_LANES": "256"}])
def test_retimed_program_reproduces_the_oracle(name, knobs, monkeypatch, tmp_path):
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]
    tape_out = em.nOutTok > 0
    nt = em.nOutTok if tape_out else em.nInTok
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    for k, v in knobs.items():
        if not v.startswith("+"): monkeypatch.setenv(k, v)
    base = capi.debug_wide_retimed(em, str(tmp_path / "base.bin"), capi.MB_VITERBI, False)
    if "MB_WIDE_RETIMED_PERIOD" in knobs: monkeypatch.setenv("MB_WIDE_RETIMED_PERIOD", str(base["period"] + int(knobs["MB_WIDE_RETIMED_PERIOD"])))
    for mode, backward in ((capi.MB_VITERBI, False), (capi.MB_FORWARD, False), (capi.MB_FORWARD, True)):
        prog = capi.debug_wide_retimed(em, str(tmp_path / "p.bin"), mode, backward)
        assert prog["S"] == em.nStates and prog["inL2"] == (1 if "MB_WIDE_GLOBAL_VECTORS" in knobs else 0)
        assert prog["lanes"] == (256 if knobs.get("MB_WIDE_LANES") == "256" or em.nStates < 192 else 1024)
        # a stream runs on into the next rotation's (the kernel's prefetch ring reads 8 slots past a period)
        assert np.array_equal(prog["tail"], prog["records"][0][:8])
        for n in (0, 1, 9, 70 if prog["lanes"] * prog["slots"] <= 4096 else 20):      # 70: beyond the 64-column token window
            seq = np.random.RandomState(n + 3).randint(1, nt + 1, size=n).astype(np.int32)
            x, y = (z, seq) if tape_out else (seq, z)
            got = simulate(prog, seq, backward, mode == capi.MB_VITERBI)
            if mode == capi.MB_VITERBI: ref = om.viterbi(x, y)
            else: ref = om.backward(x, y, oracle.SUM_EXACT) if backward else om.forward(x, y, oracle.SUM_EXACT)
            ref = ref.reshape(n + 1, em.nStates)
            if mode == capi.MB_VITERBI: assert np.array_equal(got, ref)
            else:
                fin = np.isfinite(ref)
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["tiny", "fn3-10", "composite-2", "random-40", "random-recogniser-25"])
@pytest.mark.parametrize("knobs", [{}, {"MB_WIDE_GLOBAL_VECTORS": "1"}, {"MB_WIDE_LANES": "256"}])
def test_retimed_traceback_codes_reproduce_the_oracle_paths(name, knobs, monkeypatch, tmp_path):
    """One traceback code per cell (round 4, `--align` on one-tape machines without the fp64 matrix; DPMatrix::traceBack,
    src/dpmatrix.defs.h:82-110), WITHOUT a device: the forward max program built for codes (candidate lists in the reference's
    enumeration order, the seed last) still yields the oracle's Viterbi cells bit for bit, the codes the kernel's bookkeeping
    would store decode -- `tbEntry[tbOff[state] + code]` -- to the oracle's path transition for transition, ties included
    (uniform weights tie most choices), and every code addresses an entry of its state."""
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]
    tape_out = em.nOutTok > 0
    nt = em.nOutTok if tape_out else em.nInTok
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    prog = capi.debug_wide_retimed(em, str(tmp_path / "tb.bin"), capi.MB_VITERBI, False, tb_codes=True)
    ent = prog["tbEntry"]; real = ent[ent != 0xFFFFFFFF]
    pos = np.sort(real >> 16)
    assert pos.size and np.all(np.diff(pos) > 0) and pos[-1] < em.nTransitions      # a transition is the candidate of one state, once (those of weight -inf and the silent self-loop on state 0 are none)
    assert np.all((real & 0x7fff) == np.asarray(em.src)[prog["inEid"][real >> 16]])
    walked = 0
    for n in (0, 1, 9, 33, 70 if prog["lanes"] * prog["slots"] <= 4096 else 20):
        seq = np.random.RandomState(n + 5).randint(1, nt + 1, size=n).astype(np.int32)
        x, y = (z, seq) if tape_out else (seq, z)
        cells, codes = simulate(prog, seq, False, True, tb=True)
        V = om.viterbi(x, y)
        assert np.array_equal(cells, V.reshape(n + 1, em.nStates))
        if V.reshape(-1)[-1] > -math.inf:
            assert np.array_equal(walk_codes(prog, codes, n), om.traceback(x, y, V))
            walked += 1
    assert walked >= 3


def test_retimed_program_of_the_config5_machine(tmp_path):
    """BASELINE config 5's machine itself (20-node fn3 profile . simple_introns . translate . dnapsw, 5 063 states): period 10
    where the machine's cycle allows 9, a ring of 3 vectors in LDS with some hundreds of relay entries, 37 columns in flight --
    and the Viterbi cells of short sequences, bit for bit, from the record streams alone."""
    from machineboss_amd import capi, algebra as A
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    from oracle import oracle
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(20)
    em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
    assert em.nStates == 5063
    prog = capi.debug_wide_retimed(em, str(tmp_path / "c5.bin"), capi.MB_VITERBI, False)
    assert prog["lanes"] == 1024 and prog["inL2"] == 0 and prog["NB"] == 3 and 9 <= prog["period"] <= 12 and prog["NVs"] > 5065
    assert (prog["NB"] * prog["NVs"] + 2 * prog["nPen"]) * 8 + 256 <= 160 * 1024       # ring + both penalty tables + token window in LDS
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    for n in (0, 4):
        seq = np.random.RandomState(50 + n).randint(1, 5, size=n).astype(np.int32)
        assert np.array_equal(simulate(prog, seq, False, True), om.viterbi(z, seq).reshape(n + 1, em.nStates))


@pytest.mark.parametrize("name", ["fn3-10", "composite-2", "random-40", "random-recogniser-25"])
@pytest.mark.parametrize("k,lanes", [(2, 256), (3, 64), (4, 1024)])
def test_k_part_programs_reproduce_the_oracle(name, k, lanes, tmp_path):
    """k workgroups per sequence (round 5, BASELINE config 5 on a chip with more CUs than sequences): the machine's graph cut along a
    topological order of its strongly connected components, every part a retimed program of its own, values crossing through an
    exchange array.  WITHOUT a device: replaying the parts' record streams gives the oracle's Viterbi cells bit for bit (and the
    traceback codes decode to the oracle's paths through the ONE-workgroup program's tables), Forward / Backward cells to 1e-12."""
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]
    tape_out = em.nOutTok > 0
    nt = em.nOutTok if tape_out else em.nInTok
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    try:
        pv = capi.debug_wide_parts(em, str(tmp_path / "pv.bin"), k, lanes, capi.MB_VITERBI, False)
    except RuntimeError as e:
        assert name.startswith("random") and "no k-part" in str(e)      # (a random machine may be one component)
        pytest.skip("the machine's graph has no cut")
    assert 2 <= len(pv["parts"]) <= k
    for mode, backward in ((capi.MB_VITERBI, False), (capi.MB_FORWARD, False), (capi.MB_FORWARD, True)):
        pp = capi.debug_wide_parts(em, str(tmp_path / "p.bin"), k, lanes, mode, backward)
        for n in (0, 1, 9, 70 if max(p["lanes"] * p["slots"] for p in pp["parts"]) <= 4096 else 20):
            seq = np.random.RandomState(n + 3).randint(1, nt + 1, size=n).astype(np.int32)
            x, y = (z, seq) if tape_out else (seq, z)
            got = simulate_parts(pp, seq, backward, mode == capi.MB_VITERBI)
            if mode == capi.MB_VITERBI: ref = om.viterbi(x, y)
            else: ref = om.backward(x, y, oracle.SUM_EXACT) if backward else om.forward(x, y, oracle.SUM_EXACT)
            ref = ref.reshape(n + 1, em.nStates)
            if mode == capi.MB_VITERBI: assert np.array_equal(got, ref)
            else:
                fin = np.isfinite(ref)
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)
    # traceback codes
    try:
        pt = capi.debug_wide_parts(em, str(tmp_path / "pt.bin"), k, lanes, capi.MB_VITERBI, False, tb_codes=True)
    except RuntimeError:
        return
    for n in (9, 33):
        seq = np.random.RandomState(n + 5).randint(1, nt + 1, size=n).astype(np.int32)
        x, y = (z, seq) if tape_out else (seq, z)
        cells, codes = simulate_parts(pt, seq, False, True, tb=True)
        V = om.viterbi(x, y)
        assert np.array_equal(cells, V.reshape(n + 1, em.nStates))
        # (two-transition candidates change the places: the codes decode with the tables of the parts' own lists -- a code in a merged
        #  block to the transition v -> x, and the walk finds v's own code next)
        tabs = {"S": em.nStates, "tbOff": pt["tbOff"], "tbEntry": pt["tbEntry"], "inEid": pt["inEid"]}
        if V.reshape(-1)[-1] > -math.inf: assert np.array_equal(walk_codes(tabs, codes, n), om.traceback(x, y, V))


# ---- the sweep GENERATED per machine (mb_wide_jit.cpp): what the source unrolls, replayed -------------------------------------------
WJ_W, WJ_W2, WJ_ADDR, WJ_PEN, WJ_DST, WJ_KQ, WJ_GX, WJ_XO, WJ_GL = range(9)


def simulate_jit(prog, seq, backward, mode_max, tb=False, shared=None, S=None):
    """One part of a generated kernel (capi.debug_wide_jit) replayed from its per-lane constant table -- and, for a streamed program, its
    packed address words -- with the data flow of the generated source: LDS as one array of doubles addressed in BYTES (penalty tables
    first, the ring behind them), a period per rotation of the ring, a round's slots folded into `(lds[addr] + (w + lds[pen])) + w2`, the
    lane groups reduced by the ladder the round's switch would pick for the wavefront (its first lane's word), every lane storing to its
    destination (the dummy entry for lanes without a node).  shared = (X, cells, codes, loglike): a part; None: the whole machine."""
    W, NB, NVs, kMax, rowLen, nPen, nImp = (prog[k] for k in ("lanes", "NB", "NVs", "kMax", "rowLen", "nPen", "nImp"))
    NPT, U, ringBase, dummy = prog["NPT"], prog["U"], prog["ringBase"], prog["dummyAddr"]
    nPenAll = nPen + nImp
    L = len(seq)
    tab = prog["table"]
    cons = {}
    k = 0
    for kind, index, cm, words in prog["fields"]:
        if words == 2: cons[(kind, index, cm)] = (tab[k].astype(np.uint64) | (tab[k + 1].astype(np.uint64) << np.uint64(32))).view(np.float64)
        else: cons[(kind, index, cm)] = tab[k].astype(np.int64)
        k += words
    assert k == prog["nWords"]
    streamed = prog["level"] >= 1
    if streamed:      # the items of a period in execution order: a round's slots, then its destination word
        order = []
        for r, R in enumerate(prog["rounds"]):
            order += [("a", R["firstSlot"] + q) for q in range(R["depth"])] + [("d", r)]
        assert len(order) <= prog["IP"] and prog["IP"] % prog["ring"] == 0
        pos = {it: i for i, it in enumerate(order)}
    lds = np.full(prog["ldsBytes"] // 8, np.nan)
    at = lambda a: (np.asarray(a) >> 3)
    lds[ringBase // 8: ringBase // 8 + NB * NVs] = -np.inf
    Sown = prog["Sloc"]
    for b in range(NB): lds[ringBase // 8 + b * NVs + Sown + 1] = 0.0
    if shared: X, cells, codes, llout = shared
    else:
        X = None; cells = np.full((L + 1, Sown), np.nan); codes = np.full((L + 1, Sown), -1, np.int64); llout = [None]
    tok_at = lambda c: (int(seq[L - c]) if backward else int(seq[c - 1])) if 1 <= c <= L else 0
    def write_pen(table, newest):      # what the first lanes write one period ahead (and the last lanes: the imports of column `newest`)
        base = table * nPenAll
        for e in range(nPen):
            kt, col = divmod(e, rowLen); c = newest - kt
            ok = col == 0 or (c == 0 if col == rowLen - 1 else (c >= 1 and tok_at(c) == col))
            lds[base + e] = 0.0 if ok else -np.inf
        if nImp:
            imp = X[newest, prog["impIdx"]] if newest <= L else np.full(nImp, -np.inf)
            assert not np.any(np.isnan(imp)), "an import is read before an earlier part wrote it"
            lds[base + nPen: base + nPen + nImp] = imp
    write_pen(0, 0)
    lanes = np.arange(W)
    for t in range(L + 1 + kMax):
        cm, pt = t % NB, t % NPT
        assert (t % U) % NB == cm and (t % U) % NPT == pt
        write_pen((t + 1) % NPT, t + 1)
        penOff = pt * nPenAll * 8
        for r, R in enumerate(prog["rounds"]):
            m = None; ssum = None; best = np.zeros(W, np.int64)
            for q in range(R["depth"]):
                j = R["firstSlot"] + q
                anyPen, anyW2 = prog["slots"][j]
                if streamed:
                    x = prog["stream"][cm][pos[("a", j)]].astype(np.int64)
                    addr, pen = x >> 14, (x & 0x3FFF)
                else:
                    addr = cons[(WJ_ADDR, j, cm)]; pen = cons.get((WJ_PEN, j, 0))
                assert np.all(addr % 8 == 0) and np.all(addr >= ringBase) and np.all(addr < dummy)
                w = cons[(WJ_W, j, 0)]
                v = lds[at(addr)]
                assert not np.any(np.isnan(v))
                if anyPen:
                    pv = lds[at(pen + penOff)]
                    assert not np.any(np.isnan(pv))
                    cand = v + (w + pv)
                else: cand = v + w
                if anyW2: cand = cand + cons[(WJ_W2, j, 0)]
                if mode_max:
                    if q == 0: m = cand.copy()
                    else:
                        if tb: best = np.where(cand > m, q, best)
                        m = np.maximum(m, cand)
                else:
                    if m is None: m = np.full(W, -np.inf); ssum = np.zeros(W)
                    new = np.maximum(m, cand)
                    with np.errstate(invalid="ignore"):
                        ssum = np.where(np.isneginf(new), 0.0, ssum * np.exp(np.where(np.isneginf(m), -np.inf, m - new)) + np.exp(np.where(np.isneginf(cand), -np.inf, cand - new)))
                    m = new
            # the ladder: uniform rounds a literal group size; else the wavefront's first lane selects (log2 of its largest group, | 8: masked)
            if R["uniform"]: gl = None; glane = np.full(W, R["gAll"])
            else:
                gl = cons[(WJ_GL, r, 0)]; glane = np.empty(W, np.int64)
                for w0 in range(0, W, 64):
                    head = int(gl[w0]); gw = 1 << (head & 7)
                    if not (R["gMask"] & gw): glane[w0:w0 + 64] = 1; continue      # (no case for it: the switch falls through, nothing is reduced)
                    glane[w0:w0 + 64] = (1 << (gl[w0:w0 + 64] & 7)) if (head & 8 or tb) else gw
                    assert np.all((1 << (gl[w0:w0 + 64] & 7)) <= gw)
            res = np.empty(W); key = np.zeros(W, np.int64)
            for l in range(W):
                g = int(glane[l]); l0 = l - l % g
                grp = slice(l0, l0 + g)
                if mode_max:
                    res[l] = np.max(m[grp])
                    if tb:
                        lg = int(g).bit_length() - 1 if R["uniform"] else int(gl[l] & 7)
                        own = (best[grp] << lg) | (lanes[grp] & ((1 << lg) - 1))
                        key[l] = int(np.min(own[m[grp] == res[l]]))
                else:
                    mx = float(np.max(m[grp]))
                    res[l] = -math.inf if mx == -math.inf else mx + math.log(float(np.sum(ssum[grp] * np.exp(np.where(np.isneginf(m[grp]), -np.inf, m[grp] - mx)))))
            if streamed:
                dw = prog["stream"][cm][pos[("d", r)]].astype(np.int64); dst = dw & 0x3FFFF; kq = dw >> 18
            else:
                dst = cons[(WJ_DST, r, cm)]; kq = cons.get((WJ_KQ, r, 0))
            real = dst != dummy
            assert np.all(dst[real] >= ringBase) and np.all(dst[real] < dummy) and len(set(dst[real].tolist())) == int(real.sum())      # one lane per entry
            lds[at(dst[real])] = res[real]
            gx, xo = cons.get((WJ_GX, r, 0)), cons.get((WJ_XO, r, 0))
            if kq is not None:
                c = (t - kq) if backward else (t - kMax + kq)
                ok = (c >= 0) & (c <= L)
                if xo is not None:
                    for l in np.nonzero((xo != 0xFFFFFFFF) & ok)[0]:
                        e = int(xo[l]) // 8
                        assert np.isnan(X[c[l], e]); X[c[l], e] = res[l]
                if gx is not None:
                    for l in np.nonzero((gx != 0xFFFFFFFF) & ok)[0]:
                        col = int(gx[l]) if tb else int(gx[l]) // 8
                        row = L - c[l] if backward else c[l]
                        assert real[l] and np.isnan(cells[row, col])      # every cell exactly once
                        cells[row, col] = res[l]
                        if tb: assert 0 <= key[l] < 256; codes[c[l], col] = key[l]
                if R["resultLane"] >= 0 and c[R["resultLane"]] == L: llout[0] = float(res[R["resultLane"]])
            else: assert gx is None and xo is None and R["resultLane"] < 0
            assert R["sync"] or r + 1 < len(prog["rounds"])
    if shared: return None
    assert not np.any(np.isnan(cells)) and llout[0] is not None
    return (cells, codes, llout[0]) if tb else (cells, llout[0])


def simulate_jit_parts(jp, seq, backward, mode_max, tb=False):
    L = len(seq); S = jp["S"]
    X = np.full((L + 1, max(jp["nExp"], 1)), np.nan)
    cells = np.full((L + 1, S), np.nan); codes = np.full((L + 1, S), -1, np.int64); ll = [None]
    for p in jp["parts"]: simulate_jit(p, seq, backward, mode_max, tb, shared=(X, cells, codes, ll))
    assert not np.any(np.isnan(cells)) and not np.any(np.isnan(X[:, :jp["nExp"]])) and ll[0] is not None
    return (cells, codes, ll[0]) if tb else (cells, ll[0])


@pytest.mark.parametrize("name", ["tiny", "fn3-10", "composite-2", "random-40", "random-recogniser-25"])
@pytest.mark.parametrize("level", [0, 1])
def test_generated_one_workgroup_program_reproduces_the_oracle(name, level, monkeypatch, tmp_path):
    """The kernel generated for a machine (mb_wide_jit.cpp), WITHOUT a device: its unrolled program -- rounds, slots,
    the per-lane constant table, the packed address words of the streamed form -- replayed with the generated source's data flow gives the
    oracle's Viterbi cells bit for bit, Forward / Backward cells to 1e-12, the log-likelihood from the result lane, and traceback codes
    that walk into the oracle's paths.  level 0: every constant in registers; 1: rotation-dependent words streamed."""
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]
    tape_out = em.nOutTok > 0
    nt = em.nOutTok if tape_out else em.nInTok
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    monkeypatch.setenv("MB_WIDE_JIT_LEVEL", str(level))
    for mode, backward in ((capi.MB_VITERBI, False), (capi.MB_FORWARD, False), (capi.MB_FORWARD, True)):
        jp = capi.debug_wide_jit(em, str(tmp_path / "j.hip"), k=1, mode=mode, backward=backward)
        part = jp["parts"][0]
        assert part["level"] == level and part["Sloc"] == em.nStates and part["ldsBytes"] <= 160 * 1024
        src = open(str(tmp_path / "j.hip")).read()
        assert "k_wide_jit" in src and src.count("part_0(") == 2
        for n in (0, 1, 9, 70 if part["lanes"] * part["nSlots"] <= 4096 else 20):
            seq = np.random.RandomState(n + 3).randint(1, nt + 1, size=n).astype(np.int32)
            x, y = (z, seq) if tape_out else (seq, z)
            got, ll = simulate_jit(part, seq, backward, mode == capi.MB_VITERBI)
            if mode == capi.MB_VITERBI: ref = om.viterbi(x, y)
            else: ref = om.backward(x, y, oracle.SUM_EXACT) if backward else om.forward(x, y, oracle.SUM_EXACT)
            ref = ref.reshape(n + 1, em.nStates)
            end = ref[0, 0] if backward else ref[n, em.nStates - 1]
            if mode == capi.MB_VITERBI: assert np.array_equal(got, ref) and ll == end
            else:
                fin = np.isfinite(ref)
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)
                assert (ll == end) or abs(ll - end) <= 1e-12 * abs(end)
    # traceback codes through the generated program, decoded with the interpreter program's tables (the candidate lists are the same)
    try:
        jt = capi.debug_wide_jit(em, str(tmp_path / "jt.hip"), k=1, mode=capi.MB_VITERBI, tb_codes=True)
        tabs = capi.debug_wide_retimed(em, str(tmp_path / "tb.bin"), capi.MB_VITERBI, False, tb_codes=True)
    except RuntimeError:
        return
    for n in (9, 33):
        seq = np.random.RandomState(n + 5).randint(1, nt + 1, size=n).astype(np.int32)
        x, y = (z, seq) if tape_out else (seq, z)
        cells, codes, ll = simulate_jit(jt["parts"][0], seq, False, True, tb=True)
        V = om.viterbi(x, y)
        assert np.array_equal(cells, V.reshape(n + 1, em.nStates))
        if V.reshape(-1)[-1] > -math.inf: assert np.array_equal(walk_codes(tabs, codes, n), om.traceback(x, y, V))


@pytest.mark.parametrize("name", ["fn3-10", "composite-2", "random-40"])
@pytest.mark.parametrize("k,lanes,level", [(2, 256, 0), (3, 64, 1), (4, 1024, 0), (2, 256, "attempt 2"), (3, 64, "attempt 3")])
def test_generated_k_part_programs_reproduce_the_oracle(name, k, lanes, level, monkeypatch, tmp_path):
    """... and the k-part form (one function per part in one kernel, values crossing through the exchange array): replayed part by part in
    workgroup order, against the oracle; the traceback codes decode with the joined tables of the parts' own candidate lists.
    "attempt n": the plans a build falls back to when the compiled kernel spills (wide_jit_plan: streamed words through a prefetch ring
    of 4 / 2 entries, the period padded to a multiple of it)."""
    if isinstance(level, str):
        monkeypatch.setenv("MB_WIDE_JIT_ATTEMPT", level.split()[1])
        maxRing = {"2": 4, "3": 2}[level.split()[1]]
        level = 1
    else:
        maxRing = 17
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]
    tape_out = em.nOutTok > 0
    nt = em.nOutTok if tape_out else em.nInTok
    om = oracle.OracleMachine(em)
    z = np.zeros(0, np.int32)
    monkeypatch.setenv("MB_WIDE_JIT_LEVEL", str(level))
    try:
        capi.debug_wide_jit(em, str(tmp_path / "j.hip"), k=k, lanes=lanes, mode=capi.MB_VITERBI)
    except RuntimeError as e:
        assert name.startswith("random") and "no k-part" in str(e)
        pytest.skip("the machine's graph has no cut")
    for mode, backward in ((capi.MB_VITERBI, False), (capi.MB_FORWARD, False), (capi.MB_FORWARD, True)):
        jp = capi.debug_wide_jit(em, str(tmp_path / "j.hip"), k=k, lanes=lanes, mode=mode, backward=backward)
        assert 2 <= len(jp["parts"]) <= k and all(p["level"] == level for p in jp["parts"])
        if level == 1: assert all(2 <= p["ring"] <= maxRing and p["IP"] % p["ring"] == 0 for p in jp["parts"])
        for n in (0, 1, 9, 70 if max(p["lanes"] * p["nSlots"] for p in jp["parts"]) <= 4096 else 20):
            seq = np.random.RandomState(n + 3).randint(1, nt + 1, size=n).astype(np.int32)
            x, y = (z, seq) if tape_out else (seq, z)
            got, ll = simulate_jit_parts(jp, seq, backward, mode == capi.MB_VITERBI)
            if mode == capi.MB_VITERBI: ref = om.viterbi(x, y)
            else: ref = om.backward(x, y, oracle.SUM_EXACT) if backward else om.forward(x, y, oracle.SUM_EXACT)
            ref = ref.reshape(n + 1, em.nStates)
            if mode == capi.MB_VITERBI: assert np.array_equal(got, ref)
            else:
                fin = np.isfinite(ref)
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)
    try:
        jt = capi.debug_wide_jit(em, str(tmp_path / "jt.hip"), k=k, lanes=lanes, mode=capi.MB_VITERBI, tb_codes=True)
        pt = capi.debug_wide_parts(em, str(tmp_path / "pt.bin"), k, lanes, capi.MB_VITERBI, False, tb_codes=True)
    except RuntimeError:
        return
    for n in (9, 33):
        seq = np.random.RandomState(n + 5).randint(1, nt + 1, size=n).astype(np.int32)
        x, y = (z, seq) if tape_out else (seq, z)
        cells, codes, ll = simulate_jit_parts(jt, seq, False, True, tb=True)
        V = om.viterbi(x, y)
        assert np.array_equal(cells, V.reshape(n + 1, em.nStates))
        tabs = {"S": em.nStates, "tbOff": pt["tbOff"], "tbEntry": pt["tbEntry"], "inEid": pt["inEid"]}
        if V.reshape(-1)[-1] > -math.inf: assert np.array_equal(walk_codes(tabs, codes, n), om.traceback(x, y, V))


def test_generated_source_compiles_without_a_device(tmp_path):
    """hiprtc needs no GPU: the source generated for a cut machine (sum semiring with the fp64 correction term, max semiring with
    traceback codes) compiles for gfx950 and keeps its constants in registers (no scratch memory)."""
    from machineboss_amd import capi
    em = _machines()["composite-2"]
    for kw in (dict(mode=capi.MB_FORWARD, acc=True), dict(mode=capi.MB_VITERBI, tb_codes=True), dict(mode=capi.MB_FORWARD, backward=True)):
        jp = capi.debug_wide_jit(em, str(tmp_path / "c.hip"), k=3, lanes=256, compile=True, **kw)
        assert len(jp["parts"]) >= 2
    capi.debug_wide_jit(em, str(tmp_path / "c1.hip"), k=1, mode=capi.MB_VITERBI, compile=True)


def test_generated_config5_backward_parts_with_the_fp64_term_are_planned_until_they_fit(tmp_path):
    """BASELINE config 5's machine, Backward with the fp64 correction term at 2 workgroups per sequence (the E-step of 64 x 50 kb as ONE
    chunk): the register estimate's plan spills 28 bytes, so the build plans again with a shorter prefetch ring -- and must end without
    scratch memory (round 6: this sweep silently ran through the interpreter, 376 ms beside the Forward sweep's 284)."""
    from machineboss_amd import capi, algebra as A
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(20)
    em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
    jp = capi.debug_wide_jit(em, str(tmp_path / "b.hip"), k=2, mode=capi.MB_FORWARD, backward=True, acc=True, compile=True)
    assert len(jp["parts"]) == 2 and all(p["level"] == 1 for p in jp["parts"])
    assert os.path.getsize(str(tmp_path / "b.hip.co")) > 0


def test_retimed_program_refuses_two_tape_machines(tmp_path):
    from machineboss_amd import capi
    em = random_machine(12, 2, 2, 5, density=1.5, silent_density=1.0, allow_inf=False)
    with pytest.raises(RuntimeError):
        capi.debug_wide_retimed(em, str(tmp_path / "never.bin"))

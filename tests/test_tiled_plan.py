"""The programs of the tiled family (machineboss_amd/csrc/mb_medium.hip: build_program, append_flat_usage -> k_medium_tile /
the run-time specialised kernels of mb_medium_jit.cpp), checked WITHOUT a device: `mb_debug_jit_source` with mode + 32 hands back
the chunk descriptors and records exactly as the kernels read them, `replay` below restates what `med_slow_supercell` does with
them -- chunk by chunk, every lane of a column folding `vector[srcOff] + w` over the slots of its round, the round's result
stored at the lane's `dstOff` of the current supercell -- and the cells it produces are compared with the oracle:
`src/forward.defs.h:23-49` (sum: levelled programs and the staged silent CLOSURE, whose composed weights are numeric),
`src/viterbi.cpp:18-43` (max: bit for bit), and for count programs the posterior usage of every transition,
`src/backward.cpp:58-87` -- the FLAT form (closure fill rounds + one usage pass of one transition per lane and slot, round 4) and
the levelled one.  What this pins: rounds and stages (a value is read only after the round that finalises it), node splits, the
closure's pair weights, record addressing by token, padding, the seed, the usage records (every transition exactly once, right
source vector, right destination, right accumulator).  The kernels' own arithmetic (fp32 correction terms, LDS staging, tiles) is
covered on the GPU (tests/test_gpu_parity.py)."""
import math
import os

import numpy as np
import pytest

from conftest import golden_path
from randmachine import random_machine, random_seq


def replay(prog, x, y, mode_max, bwd=None, ll=None, n_trans=0, edges=False):
    """cells[i][o][0:S] of one pair from the program (med_slow_supercell's data flow); count programs also return the usage sums.
    edges (max programs): also, per cell and state, the transition of its FIRST maximal candidate in slot order -- through the parts
    of a split node: the first maximal part's first maximal candidate -- i.e. what a traceback over this program's order chooses."""
    S, Spad, LPG, nOut, seedOff, dummy = (prog[k] for k in ("S", "Spad", "LPG", "nOut", "seedOff", "dummyOff"))
    desc, rec, flat = prog["desc"], prog["rec"], prog["flat"]
    counting, flatc = bool(prog["counting"]), bool(prog["flatCount"])
    levelled_counts = counting and not flatc
    nI, nO = len(x), len(y)
    cells = np.full((nI + 1, nO + 1, Spad), -np.inf)
    outside = np.full(Spad, -np.inf)
    lanes = np.arange(LPG)
    acc = np.zeros(n_trans + LPG)              # levelled form: one table; flat form: the after-the-loop table (register sums of loop-invariant records)
    acc_loop = np.zeros(len(prog.get("accMap", ())) + LPG)      # flat form: the loop-time table (token-selected records), entry -> transition in accMap
    two = bool(prog.get("twoTables", 0))
    wref = prog["wref"]
    # fused emit slots of the fill rounds (round 5): record -> placement (2 = VGPRs); such a record's upper srcOff half is its accumulator
    # offset, and slot 0's upper dstOff half the real state's place in the Backward supercell
    ntokT = ((prog["nIn"] + 1) * (nOut + 1), prog["nIn"] + 1, nOut + 1, 1)
    fused_place = {}
    for T, b0, place in prog.get("fused", ()):
        for k in range(ntokT[int(T)] * LPG): fused_place[int(b0) + k] = int(place)
    def add_usage(place, off_bytes, val):
        tab = acc if (not two or place == 2) else acc_loop
        np.add.at(tab, off_bytes >> 3, val)
    chosen = np.full((nI + 1, nO + 1, Spad), -1, np.int64)
    for i in range(nI + 1):
        for o in range(nO + 1):
            cur = cells[i, o]
            vecs = (cells[i - 1, o - 1] if i and o else outside, cells[i - 1, o] if i else outside, cells[i, o - 1] if o else outside, cur)
            it, ot = (int(x[i - 1]) if i else 0), (int(y[o - 1]) if o else 0)
            if counting:
                bl = np.full(Spad, -np.inf); bl[:S] = bwd[o, i] - ll      # (the oracle's matrices are [output][input][state])
            accM = np.full(LPG, -np.inf); accS = np.zeros(LPG); dst = np.zeros(LPG, np.int64)
            for dp in desc:
                hdr = int(dp[0]); ns, first, last = hdr & 15, (hdr >> 4) & 1, (hdr >> 5) & 1
                vec = vecs[(int(dp[3]) >> 24) & 255]
                base = int(dp[1]) + it * int(dp[2]) + ot * (int(dp[3]) & 0xFFFFFF) + lanes
                for k in range(ns):
                    r = rec[base + k * int(dp[4])]
                    if k == 0:
                        dst = r["dstOff"].astype(np.int64)
                        bdst = dst >> 16                      # (count programs with fused emit usage: the real state, in bytes)
                        if counting: dst = dst & 0xFFFF
                        if first:
                            seed = (dst == seedOff) if (i == 0 and o == 0) else np.zeros(LPG, bool)
                            accM = np.where(seed, 0.0, -np.inf); accS = np.where(seed, 1.0, 0.0)
                    so = r["srcOff"].astype(np.int64)
                    v = vec[(so & 0xFFFF if counting else so) >> 3] + r["w"]
                    if edges:
                        ridx = base + k * int(dp[4])
                        if k == 0 and first: best_rec = np.full(LPG, -1, np.int64)
                        better = v > accM                               # strict: the first maximum (accM is the maximum so far; -inf never wins)
                        best_rec = np.where(better, ridx, best_rec)
                    if levelled_counts:      # usage of the candidate's transition: exp(v + B(dst) - LL) into the accumulator its record names
                        with np.errstate(invalid="ignore"):
                            t = v + bl[np.minimum(dst >> 3, Spad - 1)]
                        np.add.at(acc, so >> 19, np.where(np.isnan(t) | np.isneginf(t), 0.0, np.exp(np.where(np.isfinite(t), t, 0.0))))
                    ridx0 = int(base[0]) + k * int(dp[4])
                    if flatc and ridx0 in fused_place:      # the emitting candidates of this round are real transitions into the lane's state: exp(v + B(state) - LL)
                        with np.errstate(invalid="ignore"):
                            t = v + bl[np.minimum(bdst >> 3, Spad - 1)]
                        add_usage(fused_place[ridx0], so >> 16, np.where(np.isnan(t) | np.isneginf(t), 0.0, np.exp(np.where(np.isfinite(t), t, 0.0))))
                    if mode_max: accM = np.maximum(accM, v)
                    else:
                        nm = np.maximum(accM, v); g = np.where(np.isneginf(nm), 0.0, nm)
                        accS = accS * np.exp(accM - g) + np.exp(v - g); accM = nm
                if last:
                    with np.errstate(divide="ignore"):
                        res = accM if mode_max else np.where(np.isneginf(accM), 0.0, accM) + np.log(accS)
                    ok = dst < 0x80000000
                    assert np.all(dst[ok] % 8 == 0) and np.all(dst[ok] < Spad * 8)
                    cur[dst[ok] >> 3] = res[ok]
                    if edges:
                        for ln in np.nonzero(ok & (best_rec >= 0))[0]:
                            rr = int(best_rec[ln]); e = int(wref[rr])
                            # a candidate without a transition of its own and a finite weight reads a PART of this state in the same cell
                            chosen[i, o, dst[ln] >> 3] = e if e >= 0 else chosen[i, o, int(rec[rr]["srcOff"]) >> 3]
            assert cur[S] == -np.inf                                   # the -inf sentinel padding candidates read is never written
            if flatc:
                for T, b0, place in flat:
                    tok = (it * (nOut + 1) + ot, it, ot, 0)[T]
                    r = rec[int(b0) + tok * LPG + lanes]
                    so = r["srcOff"].astype(np.int64)
                    t = (vecs[T][(so & 0xFFFF) >> 3] + r["w"]) + bl[so >> 19]
                    add_usage(int(place), r["dstOff"].astype(np.int64), np.where(np.isneginf(t), 0.0, np.exp(np.where(np.isfinite(t), t, 0.0))))
    cells = cells[:, :, :S].transpose(1, 0, 2)      # [output][input][state] like the oracle's
    if edges: return cells, chosen[:, :, :S]
    if counting and two:      # the kernel's flush: the after-the-loop table entry e, plus the loop-time table through accMap
        tot = acc[:n_trans].copy()
        np.add.at(tot, np.asarray(prog["accMap"], np.int64), acc_loop[:len(prog["accMap"])])
        return cells, tot
    return (cells, acc[:n_trans]) if counting else cells


def walk_chosen(em, chosen, x, y):
    """DPMatrix::traceBack over the transitions `replay(..., edges=True)` chose: global edge ids, start -> end."""
    src, it, ot = np.asarray(em.src), np.asarray(em.inTok), np.asarray(em.outTok)
    i, o, s, path = len(x), len(y), em.nStates - 1, []
    while i > 0 or o > 0 or s != 0:
        e = int(chosen[i, o, s]); assert e >= 0
        path.append(e); s = int(src[e]); i -= int(it[e] != 0); o -= int(ot[e] != 0)
        assert i >= 0 and o >= 0 and len(path) <= (len(x) + 1) * (len(y) + 1) * em.nStates
    return np.asarray(path[::-1], np.int64)


def _machines():
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd import algebra as A
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    return {"psw2dna": lambda: EvaluatedMachine.fromMachine(P("psw2dna"), None, useDefaults=True),
            "c4b": lambda: EvaluatedMachine.fromMachine(A.config4bMachine(os.path.dirname(golden_path("preset", "psw2dna.json"))), None, useDefaults=True),
            "random-40": lambda: random_machine(40, 2, 3, 7, density=2.0, silent_density=1.5),
            "random-100-dense": lambda: random_machine(100, 1, 2, 46019, density=2.5, silent_density=1.5),
            "random-33-inf": lambda: random_machine(33, 3, 1, 21, density=1.5, silent_density=2.0, allow_inf=True)}


def _pairs(em, rng, shapes):
    out = []
    for il, ol in shapes:
        x = random_seq(rng, il, em.nInTok); y = random_seq(rng, ol, 3 if em.nOutTok == 4 and em.nStates == 482 else em.nOutTok)      # (c4b: DNA without the stop codons' third letter, as the bench has it)
        out.append((x, y))
    return out


@pytest.mark.parametrize("name", ["psw2dna", "c4b", "random-40", "random-100-dense", "random-33-inf"])
def test_fill_programs_reproduce_the_oracle(name, tmp_path):
    """Levelled programs in both semirings (Viterbi bit for bit) and the silent closure in 2, 5 and 12 stages, at 2 and 16 columns per
    wavefront (32 and 4 lanes per column: different rounds, different node splits)."""
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]()
    om = oracle.OracleMachine(em)
    big = em.nStates > 200
    pairs = _pairs(em, np.random.RandomState(3), ((2, 5), (0, 3), (3, 0)) if big else ((4, 7), (0, 5), (6, 0), (5, 5)))
    ran = 0
    for G in (2, 16):
        for mode, closure in ((capi.MB_VITERBI, 0), (capi.MB_FORWARD, 0), (capi.MB_FORWARD, 2), (capi.MB_FORWARD, 5), (capi.MB_FORWARD, 12)):
            if big and G == 16 and closure in (0, 5): continue      # (keeps the big machines to a few seconds)
            try:
                prog = capi.debug_medium_program(em, str(tmp_path / "p.bin"), mode=mode, closure=closure, G=G)
            except capi.MbError as e:      # 16 columns of a 482-state machine do not fit the LDS
                assert "does not fit" in str(e) and big and G == 16
                continue
            ran += 1
            assert prog["S"] == em.nStates and prog["LPG"] == 64 // G and not prog["counting"]
            for x, y in pairs:
                if mode == capi.MB_VITERBI:      # cells bit for bit, and the path the program's candidate order chooses is the reference's (first maximum, ties included)
                    got, chosen = replay(prog, x, y, True, edges=True)
                    V = om.viterbi(x, y)
                    assert np.array_equal(got, V), (name, G)
                    if V.reshape(-1)[-1] > -math.inf: assert np.array_equal(walk_chosen(em, chosen, x, y), om.traceback(x, y, V)), (name, G)
                    continue
                got = replay(prog, x, y, False)
                ref = om.forward(x, y, oracle.SUM_EXACT); fin = np.isfinite(ref)
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-11, atol=1e-11), (name, G, closure)
    assert ran >= 5
    # Backward programs (the transposed machine in the reversed frame: cell (i', o') of the sweep is BackwardMatrix cell
    # (inLen - i', outLen - o'), src/backward.cpp:18-46), levelled and closure
    for closure in (0, 3):
        prog = capi.debug_medium_program(em, str(tmp_path / "b.bin"), mode=capi.MB_FORWARD, backward=True, closure=closure, G=2)
        assert prog["backward"] == 1
        for x, y in pairs:
            got = replay(prog, x[::-1], y[::-1], False)[::-1, ::-1]
            ref = om.backward(x, y, oracle.SUM_EXACT); fin = np.isfinite(ref)
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-11, atol=1e-11), (name, "backward", closure)


@pytest.mark.parametrize("name", ["psw2dna", "c4b", "random-40", "random-100-dense", "random-33-inf"])
@pytest.mark.parametrize("flat", ["1", "0"])
def test_count_programs_reproduce_the_oracle(name, flat, tmp_path, monkeypatch):
    """Both count programs against MachineCounts (src/backward.cpp:58-87, src/counts.cpp:57-64): the Forward cells of the fill rounds
    and the posterior usage of EVERY transition; in the flat form the usage records name each live transition exactly once."""
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]()
    om = oracle.OracleMachine(em)
    monkeypatch.setenv("MB_MEDIUM_COUNT_FLAT", flat)
    big = em.nStates > 200
    pairs = _pairs(em, np.random.RandomState(5), ((2, 6), (3, 4)) if big else ((4, 7), (6, 3), (0, 5), (5, 0), (3, 3)))
    pairs = [(x, y) for x, y in pairs if om.loglike(x, y, oracle.SUM_EXACT) > -math.inf]
    assert pairs
    combos = ((2, 2), (16, 6)) if flat == "1" else ((2, 0), (8, 0))
    if em.nStates > 400: combos = ((1, 6),) if flat == "1" else ((1, 0),)      # (482 states: one column per wavefront is what fits beside the accumulators)
    elif big: combos = combos[:1]
    for G, closure in combos:
        prog = capi.debug_medium_program(em, str(tmp_path / "c.bin"), mode=3, closure=closure, G=G)
        assert prog["counting"] and prog["flatCount"] == int(flat) and (prog["nFlat"] > 0) == (flat == "1")
        if flat == "1":      # every transition that is a candidate of some cell sits in exactly one usage record
            named = []
            for T, b0, place in list(prog["flat"]) + list(prog["fused"]):      # the usage pass, and the emit slots of the fill rounds that carry usage terms
                ntok = ((em.nInTok + 1) * (em.nOutTok + 1), em.nInTok + 1, em.nOutTok + 1, 1)[T]
                w = prog["wref"][int(b0):int(b0) + ntok * prog["LPG"]]
                named.append(w[(w >= 0) & np.isfinite(prog["rec"]["w"][int(b0):int(b0) + ntok * prog["LPG"]])].astype(np.int64))
            named = np.sort(np.concatenate(named))
            if not prog["fusedEmit"]: assert len(prog["fused"]) == 0
            if prog["twoTables"]:      # loop-time entries: the transitions of the usage records that are not held in VGPRs, each once
                am = np.asarray(prog["accMap"]); assert len(np.unique(am)) == len(am)
            live = np.isfinite(np.asarray(em.logWeight)) & ~((np.asarray(em.inTok) == 0) & (np.asarray(em.outTok) == 0) & (np.asarray(em.dst) <= np.asarray(em.src)))
            assert np.all(np.diff(named) > 0) and np.array_equal(named, np.nonzero(live)[0])
        ref = np.zeros(em.nTransitions); got = np.zeros(em.nTransitions)
        for x, y in pairs:
            ll = om.counts_add(x, y, ref, oracle.SUM_EXACT)
            cells, acc = replay(prog, x, y, False, bwd=om.backward(x, y, oracle.SUM_EXACT), ll=ll, n_trans=em.nTransitions)
            got += acc
            fwd = om.forward(x, y, oracle.SUM_EXACT); fin = np.isfinite(fwd)
            assert np.array_equal(np.isneginf(cells), np.isneginf(fwd)) and np.allclose(cells[fin], fwd[fin], rtol=1e-11, atol=1e-11)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (name, G, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("seed", range(10))
def test_planner_fuzz_without_a_device(seed, tmp_path, monkeypatch):
    """Random machines under random planner knobs (node splits at small degrees, part sizes, closure stage counts and cost weights,
    1 ... 32 columns per wavefront): the closure fill program, the Viterbi program and the flat count program each reproduce the
    oracle.  The device-side twin of this sweep is scripts/fuzz_knobs.sh; this one needs no GPU and runs with the CPU suite."""
    from machineboss_amd import capi
    from oracle import oracle
    rng = np.random.RandomState(1000 + seed)
    S = int(rng.choice([17, 24, 40, 64, 100, 150]))
    nIn, nOut = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    em = random_machine(S, nIn, nOut, 2000 + seed, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.3, 2.5)), allow_inf=bool(seed % 3 == 0))
    knobs = {"MB_MEDIUM_SPLIT_DEGREE": str(int(rng.choice([3, 5, 8, 12, 100000]))), "MB_MEDIUM_SPLIT_PART": str(int(rng.choice([2, 4, 8]))),
             "MB_MEDIUM_SYNC_COST": str(int(rng.choice([1, 6]))), "MB_MEDIUM_ROUND_COST": str(int(rng.choice([0, 1, 3])))}
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    G = int(rng.choice([1, 2, 4, 8, 16, 32])); K = int(rng.choice([1, 2, 3, 4, 7, 11]))
    om = oracle.OracleMachine(em)
    pairs = [(random_seq(rng, int(rng.randint(0, 7)), nIn), random_seq(rng, int(rng.randint(0, 9)), nOut)) for _ in range(3)]
    progF = capi.debug_medium_program(em, str(tmp_path / "f.bin"), mode=capi.MB_FORWARD, closure=K, G=G)
    progV = capi.debug_medium_program(em, str(tmp_path / "v.bin"), mode=capi.MB_VITERBI, closure=0, G=G)
    try:
        progC = capi.debug_medium_program(em, str(tmp_path / "c.bin"), mode=3, closure=K, G=G)
    except capi.MbError as e:
        assert "does not qualify" in str(e)
        progC = None
    ref_c = np.zeros(em.nTransitions); got_c = np.zeros(em.nTransitions)
    for x, y in pairs:
        V = om.viterbi(x, y)
        gotV, chosen = replay(progV, x, y, True, edges=True)
        assert np.array_equal(gotV, V), (knobs, G)
        if V.reshape(-1)[-1] > -math.inf: assert np.array_equal(walk_chosen(em, chosen, x, y), om.traceback(x, y, V)), (knobs, G)
        F = om.forward(x, y, oracle.SUM_EXACT); fin = np.isfinite(F)
        got = replay(progF, x, y, False)
        assert np.array_equal(np.isneginf(got), np.isneginf(F)) and np.allclose(got[fin], F[fin], rtol=1e-10, atol=1e-10), (knobs, G, K)
        if progC is not None and F.reshape(-1)[-1] > -math.inf:
            ll = om.counts_add(x, y, ref_c, oracle.SUM_EXACT)
            got_c += replay(progC, x, y, False, bwd=om.backward(x, y, oracle.SUM_EXACT), ll=ll, n_trans=em.nTransitions)[1]
    assert np.allclose(got_c, ref_c, rtol=1e-8, atol=1e-11), (knobs, G, K, float(np.abs(got_c - ref_c).max()))

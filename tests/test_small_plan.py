"""The program of the small-machine family (machineboss_amd/csrc/mb_small.cpp: small_build_host -> the generated sweeps), checked
WITHOUT a device: `mb_debug_small_source` with mode + 32 hands back what the generator unrolls -- the evaluation order of the states,
every state's candidates in the reference's enumeration order, the weight and edge-id tables with their token layout -- and `replay`
below evaluates a lattice with it the way the generated step does (candidate = neighbour cell + table weight, one fold per state,
the seed at the origin): Forward and Backward cells (`src/forward.defs.h:23-49`, `src/backward.cpp:18-46`), Viterbi cells bit for
bit (`src/viterbi.cpp:18-43`), the traceback byte of a cell = index of its first maximal candidate, decoded through the edge-id
table into the reference's path (`src/dpmatrix.defs.h:82-110`, ties included), and the posterior usage of every transition
(`src/backward.cpp:58-87`).  The kernels' own arithmetic, lane shifts and tile-major matrices are covered on the GPU."""
import math

import numpy as np
import pytest

from conftest import golden_path
from randmachine import random_machine, random_seq


def _widx(prog, T, tab, it, ot):
    nIn, nOut, off = prog["nIn"], prog["nOut"], prog["off"]
    if T == 3: return off[3] + tab
    if T == 1: return off[1] + tab * (nIn + 1) + it
    if T == 2: return off[2] + tab * (nOut + 1) + ot
    return off[0] + tab * (nIn + 1) * (nOut + 1) + it * (nOut + 1) + ot


def replay(prog, x, y, mode, bwd=None, ll=None, n_trans=0):
    """mode "sum" / "max": cells [output][input][state]; "tb": (cells, index of the first maximal candidate per cell);
    "count": (cells, usage per transition)."""
    S, nI, nO = prog["S"], len(x), len(y)
    cells = np.full((nI + 1, nO + 1, S), -np.inf)
    first = np.full((nI + 1, nO + 1, S), -1, np.int64)
    acc = np.zeros(n_trans)
    outside = np.full(S, -np.inf)
    for i in range(nI + 1):
        for o in range(nO + 1):
            cur = cells[i, o]
            vecs = (cells[i - 1, o - 1] if i and o else outside, cells[i - 1, o] if i else outside, cells[i, o - 1] if o else outside, cur)
            it, ot = (int(x[i - 1]) if i else 0), (int(y[o - 1]) if o else 0)
            for d in prog["order"]:
                cs = prog["cand"][prog["decOff"][d]:prog["decOff"][d + 1]]
                v = np.array([vecs[T][src] + prog["w"][_widx(prog, T, tab, it, ot)] for T, src, dup, tab in cs]) if len(cs) else np.zeros(0)
                if mode in ("max", "tb"):
                    res = float(np.max(v)) if v.size else -math.inf
                    if v.size: first[i, o, d] = int(np.argmax(v))      # (argmax: the first maximum, like std::max_element)
                else:
                    mx = float(np.max(v)) if v.size else -math.inf
                    res = -math.inf if mx == -math.inf else mx + math.log(float(np.sum(np.exp(v - mx))))
                if d == prog["seedState"] and i == 0 and o == 0: res = 0.0
                cur[d] = res
                if mode == "count":
                    bl = bwd[o, i, d] - ll
                    for (T, src, dup, tab), vk in zip(cs, v):
                        e = int(prog["eid"][_widx(prog, T, tab, it, ot)])
                        if e >= 0 and vk > -math.inf and bl > -math.inf: acc[e] += math.exp(vk + bl)
    cells = cells.transpose(1, 0, 2)
    return (cells, first) if mode == "tb" else ((cells, acc) if mode == "count" else cells)


def walk(prog, first, x, y):
    """k_small_traceback restated: from (inLen, outLen, end state) through the decode entries to the origin; global edge ids, start -> end."""
    i, o, s, path = len(x), len(y), prog["endState"], []
    while i > 0 or o > 0 or s != prog["seedState"]:
        k = int(first[i, o, s]); assert k >= 0
        T, src, dup, tab = (int(v) for v in prog["cand"][prog["decOff"][s] + k])
        e = int(prog["eid"][_widx(prog, T, tab, int(x[i - 1]) if i else 0, int(y[o - 1]) if o else 0)])
        assert e >= 0
        path.append(e); s = src
        if T in (0, 1): i -= 1
        if T in (0, 2): o -= 1
        assert i >= 0 and o >= 0 and len(path) <= (len(x) + 1) * (len(y) + 1) * prog["S"]
    return np.asarray(path[::-1], np.int64)


def _machines():
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    P = lambda n: EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", n + ".json")), None, useDefaults=True)
    return {"protpsw": lambda: P("protpsw"), "dnapsw": lambda: P("dnapsw"),
            "random-12": lambda: random_machine(12, 2, 3, 5, density=2.0, silent_density=1.2),
            "random-16-inf": lambda: random_machine(16, 3, 2, 9, density=1.5, silent_density=1.5, allow_inf=True),
            "random-5": lambda: random_machine(5, 1, 1, 2, density=2.5, silent_density=0.8)}


@pytest.mark.parametrize("name", ["protpsw", "dnapsw", "random-12", "random-16-inf", "random-5"])
def test_small_program_reproduces_the_oracle(name, tmp_path):
    from machineboss_amd import capi
    from oracle import oracle
    em = _machines()[name]()
    om = oracle.OracleMachine(em)
    try:
        prog = capi.debug_small_program(em, str(tmp_path / "f.bin"))
        progB = capi.debug_small_program(em, str(tmp_path / "b.bin"), backward=True)
    except capi.MbError as e:      # a random machine the family does not take (a start state fed by silent transitions ...)
        assert "does not qualify" in str(e) and name.startswith("random")
        pytest.skip(str(e))
    assert prog["S"] == em.nStates and prog["seedState"] == 0 and progB["seedState"] == em.nStates - 1
    live = prog["eid"][prog["eid"] >= 0]
    assert np.all(np.diff(np.sort(live)) > 0)                              # an edge sits in one table entry
    rng = np.random.RandomState(4)
    pairs = [(random_seq(rng, il, em.nInTok), random_seq(rng, ol, em.nOutTok)) for il, ol in ((5, 8), (0, 4), (6, 0), (7, 7), (1, 1))]
    ref_c = np.zeros(em.nTransitions); got_c = np.zeros(em.nTransitions); walked = 0
    for x, y in pairs:
        V = om.viterbi(x, y)
        cells, first = replay(prog, x, y, "tb")
        assert np.array_equal(cells, V)
        if V.reshape(-1)[-1] > -math.inf:
            assert np.array_equal(walk(prog, first, x, y), om.traceback(x, y, V)); walked += 1
        F = om.forward(x, y, oracle.SUM_EXACT); B = om.backward(x, y, oracle.SUM_EXACT)
        for got, ref in ((replay(prog, x, y, "sum"), F), (replay(progB, x[::-1], y[::-1], "sum")[::-1, ::-1], B)):
            fin = np.isfinite(ref)
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)
        if F.reshape(-1)[-1] > -math.inf:
            ll = om.counts_add(x, y, ref_c, oracle.SUM_EXACT)
            got_c += replay(prog, x, y, "count", bwd=B, ll=ll, n_trans=em.nTransitions)[1]
    assert walked >= 2 and np.allclose(got_c, ref_c, rtol=1e-10, atol=1e-13)

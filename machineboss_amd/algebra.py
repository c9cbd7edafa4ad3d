"""Transducer composition -- the caller that ASSEMBLES the machines the DP path runs on (`boss A.json B.json ...`).

Restates /root/reference/src/machine.cpp: ``Machine::compose`` (:794-907) and everything it pulls in --
``waitingMachine`` (:1053-1099), ``ergodicMachine`` (:996-1051), ``accessibleStates`` (:955-994), ``advanceSort``
(:1245-1378) with ``padWithNullStates`` / ``concatenate`` (:1380-1415,1748-1765), ``advancingMachine`` /
``updateFwdTrans`` (:1101-1230, summing silent cycles) and ``dropSilentBackTransitions`` (:1147-1175),
``TransAccumulator`` (:1923-1952), ``Machine::import`` with the consistency checks of ``Params::combine``
(src/params.cpp:19-33) and ``Constraints::combine`` (src/constraints.cpp:79-120) -- and the constructors of
``WeightAlgebra`` that composition uses (src/weight.cpp:109-190): weights stay SYMBOLIC JSON expressions, simplified
exactly as the reference simplifies them, so a composed machine evaluates to the same doubles (and therefore the
same bit-exact Viterbi matrices) as the reference's composition.

State order matters: the DP engine's silent-transition levels and the Viterbi tie-breaks depend on it, and
``advanceSort`` is a heuristic whose result is pinned by the reference's expected machines (tests/golden/expect/
bitecho-bitecho.json, bitecho-bitstutter.json, bitstutter-bitstutter.json, bitnoise-bitnoise.json,
unitindel-unitindel.json).
"""
from __future__ import annotations

import json
from typing import Any, Callable, Dict, List, Optional, Tuple

from .machine import Constraints, Machine, MachineError, MachineState, MachineTransition

MachineWaitTag, MachineCatLeftTag, MachineCatRightTag = "wait", "concat-l", "concat-r"
SumSilentCycles, BreakSilentCycles, LeaveSilentCycles = 0, 1, 2


# ---- WeightAlgebra on JSON expressions (src/weight.cpp:109-190) ------------------------------------------------------
def _isNumber(w: Any) -> bool:
    return isinstance(w, (int, float)) and not isinstance(w, bool)


def wIsZero(w: Any) -> bool:
    return w is None or w is False or (_isNumber(w) and w == 0)


def wIsOne(w: Any) -> bool:
    return w is True or (_isNumber(w) and w == 1)


def _canon(w: Any) -> Any:
    """{"not":x} is Sub(1,x) and {"geomsum":x} is Div(1,Sub(1,x)) inside the reference (src/weight.cpp:572-575)."""
    if isinstance(w, dict) and w:
        op, a = next(iter(w.items()))
        if op == "not":
            return {"-": [1, _canon(a)]}
        if op == "geomsum":
            return {"/": [1, {"-": [1, _canon(a)]}]}
        if isinstance(a, list):
            return {op: [_canon(x) for x in a]}
        return {op: _canon(a)}
    if w is True:
        return 1
    if w is False or w is None:
        return 0
    return w


def wSubtract(l: Any, r: Any) -> Any:
    return l if wIsZero(r) else {"-": [l, r]}


def wNegate(p: Any) -> Any:
    return wSubtract(1, p)


def wDivide(l: Any, r: Any) -> Any:
    if wIsOne(r):
        return l
    if wIsZero(l):
        return 0
    if _isNumber(l) and _isNumber(r) and (isinstance(l, float) or isinstance(r, float)):
        return float(l) / float(r)
    return {"/": [l, r]}


def wGeometricSum(p: Any) -> Any:
    return wDivide(1, wNegate(p))


def wMultiply(l: Any, r: Any) -> Any:
    if wIsOne(l):
        return r
    if wIsOne(r):
        return l
    if wIsZero(l) or wIsZero(r):
        return 0
    if isinstance(l, int) and isinstance(r, int) and not isinstance(l, bool) and not isinstance(r, bool):
        return l * r
    if _isNumber(l) and _isNumber(r):
        return float(l) * float(r)
    return {"*": [l, r]}


def wAdd(l: Any, r: Any) -> Any:
    if wIsZero(l):
        return r
    if wIsZero(r):
        return l
    if isinstance(r, dict) and "-" in r and wIsZero(r["-"][0]):
        return wSubtract(l, r["-"][1])
    if isinstance(l, int) and isinstance(r, int) and not isinstance(l, bool) and not isinstance(r, bool):
        return l + r
    if _isNumber(l) and _isNumber(r):
        return float(l) + float(r)
    return {"+": [l, r]}


def weightToJson(w: Any) -> Any:
    """WeightAlgebra::toJsonStream's shorthand (src/weight.cpp:464-540): Sub(1,x) prints as not, Div(1,Sub(1,x)) as geomsum."""
    if isinstance(w, dict) and w:
        op, a = next(iter(w.items()))
        if op == "-" and wIsOne(a[0]):
            return {"not": weightToJson(a[1])}
        if op == "/" and wIsOne(a[0]) and isinstance(a[1], dict) and "-" in a[1] and wIsOne(a[1]["-"][0]):
            return {"geomsum": weightToJson(a[1]["-"][1])}
        if isinstance(a, list):
            return {op: [weightToJson(x) for x in a]}
        return {op: weightToJson(a)}
    return w


# ---- MachineState predicates (src/machine.cpp:68-131) ----------------------------------------------------------------
def _terminates(ms: MachineState) -> bool: return not ms.trans
def _exitsWithInput(ms: MachineState) -> bool: return any(t.inp for t in ms.trans)
def _exitsWithoutInput(ms: MachineState) -> bool: return any(not t.inp for t in ms.trans)
def _waits(ms: MachineState) -> bool: return not _exitsWithoutInput(ms)
def _continues(ms: MachineState) -> bool: return not _exitsWithInput(ms) and not _terminates(ms)
def _exitsWithoutIO(ms: MachineState) -> bool: return any(not t.inp and not t.out for t in ms.trans)


def _copyTrans(t: MachineTransition, dest: Optional[int] = None) -> MachineTransition:
    return MachineTransition(dest=t.dest if dest is None else dest, inp=t.inp, out=t.out, weight=t.weight)


def _copyMachine(m: Machine) -> Machine:
    out = Machine(funcs=dict(m.funcs), cons=Constraints(list(m.cons.prob), [list(g) for g in m.cons.norm], list(m.cons.rate)))
    for ms in m.state:
        ns = MachineState(); ns.name = ms.name; ns.trans = [_copyTrans(t) for t in ms.trans]
        out.state.append(ns)
    return out


# ---- Machine::import (src/machine.cpp:2012-2020) ------------------------------------------------------------------------
def _normText(g: List[str]) -> str:
    return "norm{" + ",".join(g) + "}"


def combineConstraints(a: Constraints, b: Constraints) -> Constraints:
    """Constraints::combine (src/constraints.cpp:101-120) with checkRedundant's consistency test (:84-89)."""
    typ: Dict[str, str] = {}
    for p in a.prob: typ[p] = "prob[%s]" % p
    for r in a.rate: typ[r] = "rate[%s]" % r
    for g in a.norm:
        for p in g: typ[p] = _normText(g)

    def redundant(p: str, t: str) -> bool:
        if p in typ and typ[p] != t:
            raise MachineError("Inconsistent constraints for %s: %s vs %s" % (p, typ[p], t))
        return p in typ
    out = Constraints(list(a.prob), [list(g) for g in a.norm], list(a.rate))
    for p in b.prob:
        if not redundant(p, "prob[%s]" % p): out.prob.append(p)
    for r in b.rate:
        if not redundant(r, "rate[%s]" % r): out.rate.append(r)
    for g in b.norm:
        red = False
        for p in g:
            red = redundant(p, _normText(g)) or red
        if not red:
            out.norm.append(list(g))
    return out


def combineDefs(a: Dict[str, Any], b: Dict[str, Any], overwrite: bool = False) -> Dict[str, Any]:
    """Params::combine (src/params.cpp:19-33)."""
    c = dict(a)
    for name, d in b.items():
        if not overwrite and name in c:
            x, y = json.dumps(weightToJson(_canon(c[name]))), json.dumps(weightToJson(_canon(d)))
            if x != y:
                raise MachineError("Inconsistent parameter definitions for %s: %s vs %s" % (name, x, y))
        else:
            c[name] = d
    return c


def _import(dst: Machine, *srcs: Machine) -> None:
    for m in srcs:
        dst.funcs = combineDefs(dst.funcs, m.funcs)
        dst.cons = combineConstraints(dst.cons, m.cons)


# ---- small constructors ---------------------------------------------------------------------------------------------------
def nullMachine() -> Machine:
    m = Machine(); m.state.append(MachineState()); return m


def zeroMachine() -> Machine:
    m = Machine(); m.state += [MachineState(), MachineState()]; return m


def generator(seq: List[str], name: str) -> Machine:
    """Machine::generator (src/machine.cpp:1667-1675): emits exactly `seq`."""
    m = Machine()
    for pos in range(len(seq) + 1):
        ms = MachineState(); ms.name = [name, pos]
        if pos < len(seq):
            ms.trans.append(MachineTransition(dest=pos + 1, inp="", out=seq[pos], weight=1))
        m.state.append(ms)
    return m


def recognizer(seq: List[str], name: str) -> Machine:
    """Machine::recognizer (src/machine.cpp:1677-1685): accepts exactly `seq`."""
    m = Machine()
    for pos in range(len(seq) + 1):
        ms = MachineState(); ms.name = [name, pos]
        if pos < len(seq):
            ms.trans.append(MachineTransition(dest=pos + 1, inp=seq[pos], out="", weight=1))
        m.state.append(ms)
    return m


def concatenate(left: Machine, right: Machine, leftTag: str = MachineCatLeftTag, rightTag: str = MachineCatRightTag) -> Machine:
    """src/machine.cpp:1748-1765."""
    if not left.state or not right.state:
        raise MachineError("Attempt to concatenate transducer with uninitialized transducer")
    m = _copyMachine(left)
    _import(m, left, right)
    for ms in m.state:
        if ms.name is not None:
            ms.name = [leftTag, ms.name]
    n = len(left.state)
    for rs in right.state:
        ns = MachineState(); ns.name = None if rs.name is None else [rightTag, rs.name]
        ns.trans = [_copyTrans(t, t.dest + n) for t in rs.trans]
        m.state.append(ns)
    m.state[n - 1].trans.append(MachineTransition(dest=n, inp="", out="", weight=1))
    return m


# ---- reachability ---------------------------------------------------------------------------------------------------------
def accessibleStates(m: Machine) -> set:
    n = len(m.state)
    fromStart = [False] * n; q = [0]; fromStart[0] = True
    while q:
        c = q.pop(0)
        for t in m.state[c].trans:
            if not fromStart[t.dest]:
                fromStart[t.dest] = True; q.append(t.dest)
    sources: List[List[int]] = [[] for _ in range(n)]
    for s, ms in enumerate(m.state):
        for t in ms.trans:
            sources[t.dest].append(s)
    toEnd = [False] * n; q = [n - 1]; toEnd[n - 1] = True
    while q:
        c = q.pop(0)
        for s in sources[c]:
            if not toEnd[s]:
                toEnd[s] = True; q.append(s)
    return {s for s in range(n) if fromStart[s] and toEnd[s]}


def isErgodicMachine(m: Machine) -> bool:
    acc = accessibleStates(m)
    return len(acc) == len(m.state) and (len(m.state) - 1) in acc


def ergodicMachine(m: Machine) -> Machine:
    """src/machine.cpp:996-1051: drop inaccessible states and collapse unit-weight silent chains."""
    if isErgodicMachine(m):
        return m
    em = Machine(); _import(em, m)
    n = len(m.state)
    acc = accessibleStates(m)
    keep = [s in acc for s in range(n)]
    if not keep[n - 1]:
        return zeroMachine()
    nullEquiv: Dict[int, int] = {}
    for s in range(n):
        if keep[s]:
            d = s
            while len(m.state[d].trans) == 1 and m.state[d].trans[0].isSilent() and wIsOne(m.state[d].trans[0].weight):
                d = m.state[d].trans[0].dest
            if d != s:
                nullEquiv[s] = d
    old2new = [0] * n; ns = 0
    for s in range(n):
        if keep[s] and s not in nullEquiv:
            old2new[s] = ns; ns += 1
    for s in range(n):
        if keep[s] and s in nullEquiv:
            old2new[s] = old2new[nullEquiv[s]]
    if not ns:
        return zeroMachine()
    for s in range(n):
        if keep[s] and s not in nullEquiv:
            st = MachineState(); st.name = m.state[s].name
            st.trans = [_copyTrans(t, old2new[t.dest]) for t in m.state[s].trans if keep[t.dest]]
            em.state.append(st)
    if not isErgodicMachine(em):
        raise MachineError("failed to create ergodic machine")
    return em


# ---- waiting machine (src/machine.cpp:1053-1099) --------------------------------------------------------------------------
def isWaitingMachine(m: Machine) -> bool:
    return all(_waits(ms) or _continues(ms) for ms in m.state)


def waitingMachine(m: Machine, waitTag: str = MachineWaitTag) -> Machine:
    if isWaitingMachine(m):
        return m
    wm = Machine(); _import(wm, m)
    newState = [MachineState() for _ in m.state]
    for ns, ms in zip(newState, m.state):
        ns.name = ms.name; ns.trans = [_copyTrans(t) for t in ms.trans]
    old2new: List[int] = [0] * len(m.state); new2old: List[int] = []
    for s, ms in enumerate(m.state):
        old2new[s] = len(new2old); new2old.append(s)
        if not _waits(ms) and not _continues(ms):
            c, w = MachineState(), MachineState()
            c.name = ms.name; w.name = {waitTag: ms.name}
            for t in ms.trans:
                (c.trans if not t.inp else w.trans).append(_copyTrans(t))
            c.trans.append(MachineTransition(dest=len(newState), inp="", out="", weight=1))
            old2new.append(len(new2old)); new2old.append(len(newState))
            newState[s] = c; newState.append(w)
    for s in new2old:
        ms = newState[s]
        for t in ms.trans:
            t.dest = old2new[t.dest]
        wm.state.append(ms)
    if not isWaitingMachine(wm):
        raise MachineError("failed to create waiting machine")
    return wm


# ---- advancing machines ---------------------------------------------------------------------------------------------------
def nSilentBackTransitions(m: Machine) -> int:
    return sum(1 for s in range(1, len(m.state)) for t in m.state[s].trans if t.isSilent() and t.dest <= s)


def hasNullPaddingStates(m: Machine) -> bool:
    if not m.state:
        return False
    if not (len(m.state[0].trans) == 1 and _exitsWithoutIO(m.state[0])):
        return False
    esi = len(m.state) - 1
    if m.state[esi].trans:
        return False
    nullToEnd = 0
    for ms in m.state:
        for t in ms.trans:
            if t.dest == 0:
                return False
            if t.dest == esi:
                if not t.isSilent():
                    return False
                nullToEnd += 1
    return nullToEnd == 1


def padWithNullStates(m: Machine) -> Machine:
    hasNullStart = bool(m.state) and len(m.state[0].trans) == 1 and _exitsWithoutIO(m.state[0])
    if hasNullStart and any(t.dest == 0 for ms in m.state for t in ms.trans):
        hasNullStart = False
    dummy = nullMachine()
    result = m if hasNullStart else concatenate(dummy, m)
    return result if hasNullPaddingStates(result) else concatenate(result, dummy)


def advanceSort(m: Machine) -> Machine:
    """src/machine.cpp:1245-1378 with countBackTransitions = nSilentBackTransitions, mustAdvance = isSilent."""
    n = len(m.state)
    before = nSilentBackTransitions(m)
    if not before:
        return m
    silIn: List[List[int]] = [[] for _ in range(n)]; silOut: List[List[int]] = [[] for _ in range(n)]
    nIn = [0] * n; nOut = [0] * n
    for s in range(1, n - 1):
        for t in m.state[s].trans:
            if t.isSilent() and t.dest != s and t.dest != n - 1 and t.dest != 0:
                silOut[s].append(t.dest); silIn[t.dest].append(s); nOut[s] += 1; nIn[t.dest] += 1
    key = lambda a: (nIn[a], nIn[a] - nOut[a], a)      # the std::set comparator
    order: List[int] = []
    queue: set = set()

    def addToOrder(s: int):
        order.append(s)
        for nx in silOut[s]:
            nIn[nx] -= 1          # membership in `queue` is unaffected; its ordering key is recomputed on extraction
        for pv in silIn[s]:
            nOut[pv] -= 1
    addToOrder(0)
    if n > 1:
        queue = set(range(1, n - 1))
        while queue:
            nxt = min(queue, key=key)
            queue.remove(nxt)
            addToOrder(nxt)
        addToOrder(n - 1)
    old2new = [0] * n
    changed = False
    for k in range(n):
        changed = changed or order[k] != k
        old2new[order[k]] = k
    if not changed:
        result = m
    else:
        result = Machine(); _import(result, m)
        for s in order:
            ns = MachineState(); ns.name = m.state[s].name
            ns.trans = [_copyTrans(t, old2new[t.dest]) for t in m.state[s].trans]
            result.state.append(ns)
    after = nSilentBackTransitions(result)
    if after >= before and changed:
        result = m
    if after and not hasNullPaddingStates(m):
        withDummy = padWithNullStates(m)
        if not hasNullPaddingStates(withDummy):
            raise MachineError("Dummy machine does not look like a dummy, triggering infinite dummification loop")
        sortedWithDummy = advanceSort(withDummy)
        if nSilentBackTransitions(sortedWithDummy) < after:
            result = sortedWithDummy
    return result


class TransAccumulator:
    """src/machine.cpp:1923-1952: sums the weights of transitions with the same (dest, in, out); ordered maps."""

    def __init__(self):
        self.t: Dict[int, Dict[str, Dict[str, Any]]] = {}

    def accumulate(self, inp: str, out: str, dest: int, w: Any):
        d = self.t.setdefault(dest, {}).setdefault(inp, {})
        d[out] = wAdd(w, d[out]) if out in d else w

    def transitions(self) -> List[MachineTransition]:
        return [MachineTransition(dest=dest, inp=inp, out=out, weight=w)
                for dest in sorted(self.t) for inp in sorted(self.t[dest]) for out, w in sorted(self.t[dest][inp].items())]


def dropSilentBackTransitions(m: Machine) -> Machine:
    if m.isAdvancingMachine():
        return m
    am = Machine(); _import(am, m)
    for s, ms in enumerate(m.state):
        ns = MachineState(); ns.name = ms.name
        ns.trans = [_copyTrans(t) for t in ms.trans if not (t.isSilent() and t.dest <= s)]
        am.state.append(ns)
    return am


def advancingMachine(m: Machine) -> Machine:
    """src/machine.cpp:1101-1137,1177-1230: eliminate backward silent transitions by summing over silent cycles."""
    if m.isAdvancingMachine():
        return m
    am = Machine(); _import(am, m)
    n = len(m.state)
    fwd: Dict[int, Dict[int, List[MachineTransition]]] = {}

    def update(i: int, newMin: int):
        if i in fwd and newMin in fwd[i]:
            return
        old: List[MachineTransition] = []
        if newMin > i:
            update(i, newMin - 1)
            old = fwd[i][newMin - 1]
        elif newMin == i:
            old = m.state[newMin].trans
        new: List[MachineTransition] = []
        for tij in old:
            if tij.inp or tij.out:
                new.append(tij)
            else:
                j = tij.dest
                if j >= newMin:
                    new.append(tij)
                else:
                    if i != j:
                        update(j, newMin)
                    for tjk in (old if i == j else fwd[j][newMin]):
                        k = tjk.dest
                        if not ((tjk.inp or tjk.out) or (k > j and (k > i or (k == i and i == newMin)))):
                            raise MachineError("oops: cycle. i=%d j=%d k=%d" % (i, j, k))
                        new.append(MachineTransition(dest=k, inp=tjk.inp, out=tjk.out, weight=wMultiply(tij.weight, tjk.weight)))
        fwd.setdefault(i, {})[newMin] = new
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * n + 1000))
    for s in range(n):
        ns = MachineState(); ns.name = m.state[s].name
        update(s, s)
        ta = TransAccumulator()
        for t in fwd[s][s]:
            ta.accumulate(t.inp, t.out, t.dest, t.weight)
        exitSelf: Any = 1
        for t in ta.transitions():
            if t.isSilent() and t.dest == s:
                exitSelf = wGeometricSum(t.weight)
            else:
                ns.trans.append(t)
        if not wIsOne(exitSelf):
            for t in ns.trans:
                t.weight = wMultiply(exitSelf, t.weight)
        fwd[s][s] = ns.trans
        am.state.append(ns)
    if not am.isAdvancingMachine():
        raise MachineError("failed to create advancing machine")
    return am


def processCycles(m: Machine, strategy: int = SumSilentCycles) -> Machine:
    if strategy == LeaveSilentCycles:
        return m
    return advancingMachine(m) if strategy == SumSilentCycles else dropSilentBackTransitions(m)


# ---- Machine::compose (src/machine.cpp:794-907) ---------------------------------------------------------------------------
def compose(first: Machine, origSecond: Machine, assignStateNames: bool = True, collapseDegenerateTransitions: bool = True,
            cycleStrategy: int = SumSilentCycles) -> Machine:
    second = origSecond if isWaitingMachine(origSecond) else waitingMachine(origSecond)
    iStates, jStates = len(first.state), len(second.state)
    allNull = lambda mm: all(ms.name is None for ms in mm.state)
    assignStateNames = assignStateNames and not allNull(first) and not allNull(second)

    def dests(c: int) -> List[Tuple[str, str, int, Any]]:
        i, j = divmod(c, jStates)
        msi, msj = first.state[i], second.state[j]
        out = []
        if _waits(msj) or _terminates(msj):
            for it in msi.trans:
                if not it.out:
                    out.append((it.inp, "", it.dest * jStates + j, it.weight))
                else:
                    for jt in msj.trans:
                        if it.out == jt.inp:
                            out.append((it.inp, jt.out, it.dest * jStates + jt.dest, wMultiply(it.weight, jt.weight)))
        else:
            for jt in msj.trans:
                out.append(("", jt.out, i * jStates + jt.dest, jt.weight))
        return out
    keep = [False] * (iStates * jStates)
    toVisit = [0]; keep[0] = True; kept: List[int] = []
    while toVisit:
        c = toVisit.pop()
        kept.append(c)
        for _, _, d, _ in dests(c):
            if not keep[d]:
                keep[d] = True; toVisit.append(d)
    if not keep[iStates * jStates - 1]:
        return zeroMachine()        # "End state of composed machine is not accessible"
    kept.sort()
    comp2kept = {c: k for k, c in enumerate(kept)}
    cm = Machine(); _import(cm, first, second)
    for c in kept:
        i, j = divmod(c, jStates)
        ms = MachineState()
        if assignStateNames:
            ms.name = [first.state[i].name, second.state[j].name]
        if collapseDegenerateTransitions:
            ta = TransAccumulator()
            for inp, out, d, w in dests(c):
                ta.accumulate(inp, out, comp2kept[d], w)
            ms.trans = ta.transitions()
        else:
            ms.trans = [MachineTransition(dest=comp2kept[d], inp=inp, out=out, weight=w) for inp, out, d, w in dests(c)]
        cm.state.append(ms)
    return ergodicMachine(processCycles(advanceSort(ergodicMachine(cm)), cycleStrategy))


def composeAll(machines: List[Machine]) -> Machine:
    """boss's implicit reduction of several machines on one command line (target/boss.cpp:268-276): right to left."""
    m = machines[-1]
    for prev in reversed(machines[:-1]):
        m = compose(prev, m, True, True, SumSilentCycles)
    return m


def composeLeftToRight(machines: List[Machine]) -> Machine:
    """((a . b) . c) . d -- pairwise Machine::compose from the left, the order SURVEY.md section 8(d) probed config 5 in
    (fn3 . simple_introns . translate . dnapsw = 21 761 states / 63 267 transitions).  Every intermediate result is trimmed
    to its accessible states, so the state count differs from the right-to-left order of the `boss` command line."""
    m = machines[0]
    for nxt in machines[1:]:
        m = compose(m, nxt, True, True, SumSilentCycles)
    return m


def machineToJson(m: Machine, showParams: bool = False) -> dict:
    """The structure Machine::writeJson prints (src/machine.cpp:203-345), for comparison with expected machines."""
    states = []
    for n, ms in enumerate(m.state):
        sj: Dict[str, Any] = {"n": n}
        if ms.name is not None:
            sj["id"] = ms.name
        tr = []
        for t in ms.trans:
            tj: Dict[str, Any] = {"to": t.dest}
            if t.inp: tj["in"] = t.inp
            if t.out: tj["out"] = t.out
            if not wIsOne(t.weight): tj["weight"] = weightToJson(_canon(t.weight))
            tr.append(tj)
        if tr:
            sj["trans"] = tr
        states.append(sj)
    out: Dict[str, Any] = {"state": states}
    if showParams:
        if m.funcs:
            out["defs"] = {k: weightToJson(_canon(v)) for k, v in m.funcs.items()}
        if not m.cons.empty():
            cj: Dict[str, Any] = {}
            if m.cons.norm: cj["norm"] = m.cons.norm
            if m.cons.prob: cj["prob"] = m.cons.prob
            if m.cons.rate: cj["rate"] = m.cons.rate
            out["cons"] = cj
    return out


def config4bMachine(presetDir: str) -> Machine:
    """BASELINE config 4 read literally ("C4b"): protpsw . translate . dnapsw with dnapsw's parameter
    constraints cleared -- with them the reference's compose aborts on the shared parameter names (src/machine.cpp:794-907).
    482 states, 3095 transitions, 22 silent levels."""
    import os
    from .machine import Constraints
    P = lambda n: Machine.fromFile(os.path.join(presetDir, n + ".json"))
    d = P("dnapsw")
    d.cons = Constraints()
    return compose(compose(P("protpsw"), P("translate")), d)

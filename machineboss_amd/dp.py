"""Host-side mirror of the reference's DP class interface, backed by the HIP engine through the C-ABI.

Same names, argument meaning and error behaviour as /root/reference/src/{forward,backward,viterbi,counts}.h:
construction is computation (src/forward.defs.h:1-21, viterbi.cpp:6-16, backward.cpp:6-16), results are read
through ``logLike()``, ``cell()``, ``path()``, ``getCounts()`` / ``MachineCounts.count``.

A ``SeqPair`` here is a pair of symbol lists (plus optional names); batches are lists of them.  All numerics are
done on the GPU -- these classes only tokenise, marshal and map edge ids back to (state, transIndex).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import capi
from .evalmachine import EvaluatedMachine
from .machine import Machine, MachineError, MachineTransition, evalWeight


@dataclass
class SeqPair:
    """src/seqpair.h:56-73 (names + symbol sequences; alignments/envelopes are not on this path yet)."""
    input: List[str]
    output: List[str]
    inputName: str = "input"
    outputName: str = "output"

    @classmethod
    def fromJson(cls, j: dict) -> "SeqPair":
        return cls(list(j["input"]["sequence"]), list(j["output"]["sequence"]),
                   j["input"].get("name", "input"), j["output"].get("name", "output"))


def _device_machine(em: EvaluatedMachine) -> capi.DeviceMachine:
    dm = getattr(em, "_device", None)
    if dm is None or dm.h is None:
        dm = capi.DeviceMachine(em)
        em._device = dm
    return dm


@dataclass
class MachinePath:
    """src/machine.h MachinePath: the transitions of a path, start to end."""
    trans: List[MachineTransition] = field(default_factory=list)
    steps: List[tuple] = field(default_factory=list)   # (srcState, transIndex) per transition


class _DPMatrix:
    """DPMatrix<IdentityIndexMapper> (src/dpmatrix.h:64-163): full matrix, cell() = -inf outside the lattice."""
    _mode = capi.MB_FORWARD

    def __init__(self, machine: EvaluatedMachine, seqPair: SeqPair, startState: int = 0):
        self.machine, self.seqPair = machine, seqPair
        self.input = machine.inputTokenizer.tokenize(seqPair.input)     # raises like Tokenizer::tokenize
        self.output = machine.outputTokenizer.tokenize(seqPair.output)
        self.inLen, self.outLen, self.nStates = len(self.input), len(self.output), machine.nStates
        self._dm = _device_machine(machine)
        self._cells = self._dm.fill(self._mode, self.input, self.output, startState)  # [o][i][s]

    def cell(self, inPos: int, outPos: int, state: int) -> float:
        if 0 <= outPos <= self.outLen and 0 <= inPos <= self.inLen:
            return float(self._cells[outPos, inPos, state])
        return -math.inf

    def startCell(self) -> float:
        return self.cell(0, 0, self.machine.startState())

    def endCell(self) -> float:
        return self.cell(self.inLen, self.outLen, self.machine.endState())

    def cells(self) -> np.ndarray:
        return self._cells


class ForwardMatrix(_DPMatrix):
    """src/forward.h:19-27."""
    _mode = capi.MB_FORWARD

    def logLike(self) -> float:
        return self.endCell()


class BackwardMatrix(_DPMatrix):
    """src/backward.h:44-59 (fill + logLike; getCounts is served by MachineCounts on the device)."""
    _mode = capi.MB_BACKWARD

    def logLike(self) -> float:
        return self.startCell()


class ViterbiMatrix(_DPMatrix):
    """src/viterbi.h:9-18."""
    _mode = capi.MB_VITERBI

    def __init__(self, machine: EvaluatedMachine, seqPair: SeqPair):
        super().__init__(machine, seqPair, 0)

    def logLike(self) -> float:
        return self.endCell()

    def path(self, m: Machine) -> MachinePath:
        """traceBack(m) (src/viterbi.cpp:49-51, dpmatrix.defs.h:61-110), run on the device."""
        if not (self.endCell() > -math.inf):
            raise MachineError("Can't do traceback: no finite-weight paths")
        b = capi.DeviceBatch.from_pairs(self._dm, [(self.input, self.output)])
        _, off, edges = b.viterbi(paths=True)
        b.close()
        return edgesToPath(self.machine, m, edges)


def edgesToPath(em: EvaluatedMachine, m: Machine, edges: Sequence[int]) -> MachinePath:
    p = MachinePath()
    for e in edges:
        s, ti = int(em.src[e]), int(em.transIndex[e])
        p.steps.append((s, ti))
        p.trans.append(m.state[s].getTransition(ti))
    return p


def forwardLogLikeBatch(machine: EvaluatedMachine, pairs: Sequence[SeqPair], rolling: bool = True) -> List[float]:
    """The `--loglike` loop of target/boss.cpp:792-808: pairs that cannot be tokenised score -inf."""
    dm = _device_machine(machine)
    ok = [machine.canTokenize(sp.input, sp.output) for sp in pairs]
    toks = [(machine.inputTokenizer.tokenize(sp.input), machine.outputTokenizer.tokenize(sp.output))
            for sp, k in zip(pairs, ok) if k]
    out = [-math.inf] * len(pairs)
    if toks:
        b = capi.DeviceBatch.from_pairs(dm, toks)
        ll = b.forward(capi.MB_ROLLING if rolling else capi.MB_MATERIALISE)
        b.close()
        it = iter(ll)
        for k, good in enumerate(ok):
            if good:
                out[k] = float(next(it))
    return out


class MachineCounts:
    """src/counts.h:11-25: E-step sufficient statistics, count[state][transIndex] and loglike."""

    def __init__(self, machine: EvaluatedMachine, seqPairs: Optional[Sequence[SeqPair]] = None):
        self.machine = machine
        self.init(machine)
        if seqPairs is not None:
            self.addBatch(seqPairs)

    def init(self, machine: EvaluatedMachine):
        self.loglike = 0.0
        self._flat = np.zeros(machine.nTransitions, np.float64)

    @property
    def count(self) -> List[List[float]]:
        off = self.machine.transOffset
        return [list(self._flat[off[s]:off[s + 1]]) for s in range(self.machine.nStates)]

    def add(self, seqPair: SeqPair) -> float:
        return self.addBatch([seqPair])[0]

    def addBatch(self, seqPairs: Sequence[SeqPair]) -> List[float]:
        dm = _device_machine(self.machine)
        toks = [(self.machine.inputTokenizer.tokenize(sp.input), self.machine.outputTokenizer.tokenize(sp.output))
                for sp in seqPairs]
        b = capi.DeviceBatch.from_pairs(dm, toks)
        _, s, ll = b.counts(self._flat)
        b.close()
        self.loglike += s
        return [float(x) for x in ll]

    def __iadd__(self, other: "MachineCounts") -> "MachineCounts":
        self._flat += other._flat
        self.loglike += other.loglike
        return self

    def paramCounts(self, m: Machine, prob: Dict[str, Any]) -> Dict[str, float]:
        """src/counts.cpp:88-106: sum over transitions of  count * (dw/dp) * p / w."""
        defs = dict(m.funcs); defs.update(prob)
        out: Dict[str, float] = {}
        e = 0
        for s, ms in enumerate(m.state):
            for t in ms.trans:
                c = float(self._flat[e]); e += 1
                params = _freeParams(t.weight, m.funcs)
                if not params:
                    continue
                w = evalWeight(t.weight, defs)
                for p in sorted(params):
                    d = _deriv(t.weight, defs, p)
                    out[p] = out.get(p, 0.0) + c * d * float(evalWeight(p, defs)) / w
        return out


def _freeParams(w: Any, funcs: Dict[str, Any]) -> set:
    out: set = set()
    if isinstance(w, str):
        if w in funcs and not isinstance(funcs[w], (int, float)):
            out |= _freeParams(funcs[w], {k: v for k, v in funcs.items() if k != w})
        elif w not in funcs:
            out.add(w)
    elif isinstance(w, dict) and w:
        _, args = next(iter(w.items()))
        for a in (args if isinstance(args, list) else [args]):
            out |= _freeParams(a, funcs)
    return out


def _deriv(w: Any, defs: Dict[str, Any], p: str) -> float:
    """d eval(w) / d p, forward-mode over the JSON expression (the reference differentiates symbolically,
    src/weight.cpp:302-380, then evaluates; the value is the same)."""
    if w is None or isinstance(w, (bool, int, float)):
        return 0.0
    if isinstance(w, str):
        if w == p:
            return 1.0
        v = defs.get(w)
        if v is None or isinstance(v, (int, float)):
            return 0.0
        return _deriv(v, {k: x for k, x in defs.items() if k != w}, p)
    op, args = next(iter(w.items()))
    ev = lambda x: evalWeight(x, defs)
    if op == "log":
        return _deriv(args, defs, p) / ev(args)
    if op == "exp":
        return _deriv(args, defs, p) * math.exp(ev(args))
    if op == "not":
        return -_deriv(args, defs, p)
    if op == "geomsum":
        return _deriv(args, defs, p) / (1.0 - ev(args)) ** 2
    a, b = args
    if op == "*":
        return _deriv(a, defs, p) * ev(b) + ev(a) * _deriv(b, defs, p)
    if op == "/":
        return (_deriv(a, defs, p) * ev(b) - ev(a) * _deriv(b, defs, p)) / ev(b) ** 2
    if op == "+":
        return _deriv(a, defs, p) + _deriv(b, defs, p)
    if op == "-":
        return _deriv(a, defs, p) - _deriv(b, defs, p)
    raise MachineError("Unknown opcode %s" % op)

"""EvaluatedMachine: tokenised alphabets + numeric log-weights, flattened for the device.

Mirrors the *contract* of /root/reference/src/eval.{h,cpp}:

* token 0 is epsilon, tokens 1..N are the sorted alphabet symbols          (src/eval.h:13-22)
* logWeight = log(eval(weight, params))                                     (src/eval.cpp:59)
* transIndex = position of the transition in its source state's list;
  transOffset = prefix sum of out-degrees; nTransitions = total             (src/eval.cpp:51-69)
* the machine must be "advancing" (silent edges go to higher states)        (src/eval.cpp:44)

Instead of the reference's nested ``map<in, map<out, multimap<state, Trans>>>`` (src/eval.h:66-68)
the flat form is struct-of-arrays over *global edge ids* ``e = transOffset[src] + transIndex``:
``src[e], dst[e], inTok[e], outTok[e], logWeight[e]``.  The iteration order the reference gets
from those maps -- (inTok, outTok, src, insertion order) for ``incoming`` and
(inTok, outTok, dst, insertion order) for ``outgoing`` -- is re-derived inside the C-ABI library
(and, independently, inside the oracle) by a stable sort, see include/mbhip.h.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .machine import Machine, MachineError, evalWeight


class Tokenizer:
    """src/eval.h:13-48."""

    def __init__(self, symbols: Sequence[str]):
        self.tok2sym: List[str] = [""] + list(symbols)
        self.sym2tok: Dict[str, int] = {s: i for i, s in enumerate(self.tok2sym)}

    @staticmethod
    def emptyToken() -> int:
        return 0

    def canTokenize(self, seq: Sequence[str]) -> bool:
        return all(s in self.sym2tok for s in seq)

    def tokenize(self, seq: Sequence[str]) -> np.ndarray:
        out = np.empty(len(seq), dtype=np.int32)
        for i, s in enumerate(seq):
            if s not in self.sym2tok:
                raise MachineError("Can't tokenize symbol %s using this alphabet: %s" % (s, " ".join(self.tok2sym)))
            out[i] = self.sym2tok[s]
        return out

    def detokenize(self, toks: Sequence[int]) -> List[str]:
        return [self.tok2sym[t] for t in toks]


@dataclass
class EvaluatedMachine:
    nStates: int
    inputTokenizer: Tokenizer
    outputTokenizer: Tokenizer
    src: np.ndarray          # uint32[nTransitions]
    dst: np.ndarray          # uint32[nTransitions]
    inTok: np.ndarray        # uint16[nTransitions]
    outTok: np.ndarray       # uint16[nTransitions]
    transIndex: np.ndarray   # uint32[nTransitions]  index within src's transition list
    logWeight: np.ndarray    # float64[nTransitions]
    transOffset: np.ndarray  # int64[nStates+1]
    stateNames: List[Any]

    @property
    def nTransitions(self) -> int:
        return int(self.src.shape[0])

    @property
    def nInTok(self) -> int:      # alphabet size, excluding epsilon
        return len(self.inputTokenizer.tok2sym) - 1

    @property
    def nOutTok(self) -> int:
        return len(self.outputTokenizer.tok2sym) - 1

    def startState(self) -> int:
        return 0

    def endState(self) -> int:
        return self.nStates - 1

    def canTokenize(self, inSeq: Sequence[str], outSeq: Sequence[str]) -> bool:
        return self.inputTokenizer.canTokenize(inSeq) and self.outputTokenizer.canTokenize(outSeq)

    @classmethod
    def fromMachine(cls, machine: Machine, params: Optional[Dict[str, Any]] = None,
                    useDefaults: bool = False) -> "EvaluatedMachine":
        """``EvaluatedMachine(machine, params)`` (src/eval.cpp:26-70).

        ``params=None`` and ``useDefaults=False`` reproduces the parameter-free constructor
        (all log-weights zero, src/eval.cpp:33-38, :59).
        """
        if not machine.isAdvancingMachine():
            raise MachineError("Machine is not topologically sorted")
        defs: Optional[Dict[str, Any]] = None
        if params is not None or useDefaults:
            defs = machine.getParamDefs(useDefaults)
            if params:
                defs.update(params)
        it = Tokenizer(machine.inputAlphabet())
        ot = Tokenizer(machine.outputAlphabet())
        nT = machine.nTransitions()
        src = np.empty(nT, np.uint32); dst = np.empty(nT, np.uint32)
        itok = np.empty(nT, np.uint16); otok = np.empty(nT, np.uint16)
        tidx = np.empty(nT, np.uint32); lw = np.empty(nT, np.float64)
        off = np.zeros(machine.nStates() + 1, np.int64)
        e = 0
        for s, ms in enumerate(machine.state):
            off[s] = e
            for ti, t in enumerate(ms.trans):
                src[e] = s; dst[e] = t.dest
                itok[e] = it.sym2tok[t.inp]; otok[e] = ot.sym2tok[t.out]
                tidx[e] = ti
                if defs is None:
                    lw[e] = 0.0
                else:
                    w = evalWeight(t.weight, defs)
                    lw[e] = math.log(w) if w > 0 else (-math.inf if w == 0 else math.nan)
                e += 1
        off[machine.nStates()] = e
        return cls(machine.nStates(), it, ot, src, dst, itok, otok, tidx, lw, off,
                   [ms.name for ms in machine.state])

    def reweighted(self, machine: Machine, params: Optional[Dict[str, Any]] = None, useDefaults: bool = False) -> "EvaluatedMachine":
        """The same topology under other parameters: ``EvaluatedMachine.fromMachine(machine, params)`` without rebuilding tokenisers and
        index arrays, the weight expressions evaluated through a program compiled once per machine (``CompiledWeights``) -- what an EM
        iteration needs (src/fitter.cpp:28-29 rebuilds the whole object).  Log-weights are the bits ``fromMachine`` gives."""
        cw = getattr(machine, "_compiledWeights", None)
        sig = hash(tuple(id(t.weight) for ms in machine.state for t in ms.trans))      # (a transition given another weight object: compile again)
        if cw is None or cw.nTransitions != self.nTransitions or cw.signature != sig:
            cw = CompiledWeights(machine)
            cw.signature = sig
            machine._compiledWeights = cw
        defs = machine.getParamDefs(useDefaults)
        if params:
            defs.update(params)
        return self.withLogWeights(cw.logWeights(defs))

    def withLogWeights(self, logWeight: np.ndarray) -> "EvaluatedMachine":
        lw = np.ascontiguousarray(logWeight, dtype=np.float64)
        assert lw.shape == self.logWeight.shape
        return EvaluatedMachine(self.nStates, self.inputTokenizer, self.outputTokenizer, self.src, self.dst,
                                self.inTok, self.outTok, self.transIndex, lw, self.transOffset, self.stateNames)

    # --- reference iteration orders (used by tests and the Python-side traceback mapping) ---
    def incomingOrder(self) -> np.ndarray:
        """Edge ids sorted as the reference iterates state[d].incoming: (dst, inTok, outTok, src, transIndex)."""
        return np.lexsort((self.transIndex, self.src, self.outTok, self.inTok, self.dst)).astype(np.int64)

    def outgoingOrder(self) -> np.ndarray:
        """Edge ids sorted as the reference iterates state[s].outgoing: (src, inTok, outTok, dst, transIndex)."""
        return np.lexsort((self.transIndex, self.dst, self.outTok, self.inTok, self.src)).astype(np.int64)

    def silentLevels(self) -> np.ndarray:
        """level[s] = 1 + max(level[src]) over silent incoming edges, 0 if none."""
        lvl = np.zeros(self.nStates, np.int32)
        sil = [e for e in range(self.nTransitions) if self.inTok[e] == 0 and self.outTok[e] == 0]
        sil.sort(key=lambda e: int(self.dst[e]))
        for e in sil:
            s, d = int(self.src[e]), int(self.dst[e])
            if s < d:
                lvl[d] = max(lvl[d], lvl[s] + 1)
        return lvl


class CompiledWeights:
    """The weight expressions of a machine's transitions (src/weight.cpp:241-300, JSON objects) as ONE flat program, built once:
    every distinct sub-expression -- by object identity: a composed machine's products share their factors, src/machine.cpp:794-907 --
    is a node; nodes are evaluated level by level, the four arithmetic opcodes, ``not`` and ``geomsum`` as numpy array operations (IEEE
    double, the operand order of the expression: the same bits as the scalar evaluation), ``log / exp / pow`` and the final logarithm
    through ``math`` one by one (a vectorised libm need not round like the scalar one).  Parameters that are themselves expressions
    (function definitions) are evaluated by ``evalWeight``.  5 063-state composition of BASELINE config 5: 14 691 transitions,
    101 ms per evaluation through ``EvaluatedMachine.fromMachine`` -> a few ms."""

    _VEC = {"*": 0, "/": 1, "+": 2, "-": 3, "not": 4, "geomsum": 5}

    def __init__(self, machine: Machine, expand: Optional[Dict[str, Any]] = None, keep: Optional[set] = None):
        """``expand``: names whose definitions (expressions) become part of the program instead of being evaluated per call -- the
        M-step's gradient runs through them; ``keep``: names that stay leaves whatever ``expand`` says (the free parameters)."""
        self.nodes: List[tuple] = []          # (kind, a, b): kind "c" constant a / "p" parameter name a / opcode with child indices
        self._byId: Dict[int, int] = {}
        self._keep: List[Any] = []            # the expression objects (ids stay valid while they live)
        self._consts: Dict[Any, int] = {}
        self._expand = dict(expand or {}); self._leaf = set(keep or ()); self._open: set = set()
        self.top: List[int] = []
        for ms in machine.state:
            for t in ms.trans:
                self.top.append(self._node(t.weight))
        self.nTransitions = len(self.top)
        n = len(self.nodes)
        depth = [0] * n
        for i, (kind, a, b) in enumerate(self.nodes):      # children precede parents by construction
            if kind not in ("c", "p"):
                depth[i] = 1 + max(depth[a], depth[b] if b is not None else 0)
        self.constIdx = np.array([i for i, nd in enumerate(self.nodes) if nd[0] == "c"], np.int64)
        self.constVal = np.array([self.nodes[i][1] for i in self.constIdx], np.float64)
        self.params = [(i, nd[1]) for i, nd in enumerate(self.nodes) if nd[0] == "p"]
        groups: Dict[tuple, List[int]] = {}
        for i, (kind, a, b) in enumerate(self.nodes):
            if kind not in ("c", "p"):
                groups.setdefault((depth[i], kind), []).append(i)
        self.steps = []
        for (d, kind) in sorted(groups, key=lambda k: k[0]):
            idx = np.array(groups[(d, kind)], np.int64)
            a = np.array([self.nodes[i][1] for i in idx], np.int64)
            b = np.array([self.nodes[i][2] if self.nodes[i][2] is not None else 0 for i in idx], np.int64)
            self.steps.append((kind, idx, a, b))
        self.topIdx = np.array(self.top, np.int64)
        self.signature = None

    def _node(self, w: Any) -> int:
        if w is None or isinstance(w, (bool, int, float)):
            v = 0.0 if w is None else (1.0 if w is True else (0.0 if w is False else float(w)))
            key = (v, math.copysign(1.0, v)) if v == v else "nan"
            if key not in self._consts:
                self._consts[key] = len(self.nodes); self.nodes.append(("c", v, None))
            return self._consts[key]
        if isinstance(w, str):
            key = ("p", w)
            if key not in self._consts:
                d = self._expand.get(w)
                if d is not None and w not in self._leaf and not isinstance(d, (bool, int, float)) and w not in self._open:
                    self._open.add(w)      # (a definition in terms of other names: part of the program)
                    self._consts[key] = self._node(d)
                    self._open.discard(w)
                else:
                    self._consts[key] = len(self.nodes); self.nodes.append(("p", w, None))
            return self._consts[key]
        if id(w) in self._byId:
            return self._byId[id(w)]
        if isinstance(w, list):
            raise MachineError("Unexpected type in WeightExpr: array")
        if not isinstance(w, dict) or len(w) == 0:
            raise MachineError("WeightExpr must be JSON object with an opcode")
        op, args = next(iter(w.items()))
        if op in ("log", "exp", "not", "geomsum"):
            nd = (op, self._node(args), None)
        elif op in ("*", "/", "+", "-", "pow"):
            nd = (op, self._node(args[0]), self._node(args[1]))
        else:
            raise MachineError("Unknown opcode %s in JSON" % op)
        self._byId[id(w)] = len(self.nodes); self._keep.append(w)
        self.nodes.append(nd)
        return self._byId[id(w)]

    def values(self, defs: Dict[str, Any]) -> np.ndarray:
        """Weight of every transition under ``defs`` (parameter values and function definitions)."""
        v = np.empty(len(self.nodes), np.float64)
        if len(self.constIdx):
            v[self.constIdx] = self.constVal
        for i, name in self.params:
            v[i] = evalWeight(name, defs)
        with np.errstate(all="ignore"):
            for kind, idx, a, b in self.steps:
                if kind == "*": v[idx] = v[a] * v[b]
                elif kind == "/":
                    if np.any(v[b] == 0.0): raise ZeroDivisionError("float division by zero")      # (as the scalar evaluation does)
                    v[idx] = v[a] / v[b]
                elif kind == "+": v[idx] = v[a] + v[b]
                elif kind == "-": v[idx] = v[a] - v[b]
                elif kind == "not": v[idx] = 1.0 - v[a]
                elif kind == "geomsum":
                    if np.any(v[a] == 1.0): raise ZeroDivisionError("float division by zero")
                    v[idx] = 1.0 / (1.0 - v[a])
                elif kind == "log":
                    for i, x in zip(idx, v[a]): v[i] = math.log(x)
                elif kind == "exp":
                    for i, x in zip(idx, v[a]): v[i] = math.exp(x)
                else:
                    for i, x, y in zip(idx, v[a], v[b]): v[i] = math.pow(x, y)
        return v[self.topIdx]

    def objective(self, counts: np.ndarray, defs: Dict[str, Any], wantGrad: bool = True):
        """E = - sum_e counts[e] log w_e (MachineObjective, src/counts.cpp:122-131) and dE/dp for every parameter leaf, by one reverse
        sweep over the program (the reference differentiates every transition's expression symbolically, src/weight.cpp deriv)."""
        v = np.empty(len(self.nodes), np.float64)
        if len(self.constIdx):
            v[self.constIdx] = self.constVal
        for i, name in self.params:
            v[i] = evalWeight(name, defs)
        with np.errstate(all="ignore"):
            for kind, idx, a, b in self.steps:
                if kind == "*": v[idx] = v[a] * v[b]
                elif kind == "/": v[idx] = v[a] / v[b]
                elif kind == "+": v[idx] = v[a] + v[b]
                elif kind == "-": v[idx] = v[a] - v[b]
                elif kind == "not": v[idx] = 1.0 - v[a]
                elif kind == "geomsum": v[idx] = 1.0 / (1.0 - v[a])
                elif kind == "log": v[idx] = np.log(v[a])
                elif kind == "exp": v[idx] = np.exp(v[a])
                else: v[idx] = np.power(v[a], v[b])
            w = v[self.topIdx]
            used = counts != 0.0
            if np.any(~(w[used] > 0.0)) or not np.all(np.isfinite(w[used])):
                return math.inf, None
            E = -float(np.sum(counts[used] * np.log(w[used])))
            if not wantGrad:
                return E, None
            bar = np.zeros(len(self.nodes), np.float64)
            np.add.at(bar, self.topIdx[used], -counts[used] / w[used])
            for kind, idx, a, b in reversed(self.steps):
                g = bar[idx]
                if kind == "*": np.add.at(bar, a, g * v[b]); np.add.at(bar, b, g * v[a])
                elif kind == "/": np.add.at(bar, a, g / v[b]); np.add.at(bar, b, -g * v[a] / (v[b] * v[b]))
                elif kind == "+": np.add.at(bar, a, g); np.add.at(bar, b, g)
                elif kind == "-": np.add.at(bar, a, g); np.add.at(bar, b, -g)
                elif kind == "not": np.add.at(bar, a, -g)
                elif kind == "geomsum": np.add.at(bar, a, g * v[idx] * v[idx])
                elif kind == "log": np.add.at(bar, a, g / v[a])
                elif kind == "exp": np.add.at(bar, a, g * v[idx])
                else: np.add.at(bar, a, g * v[b] * np.power(v[a], v[b] - 1.0)); np.add.at(bar, b, g * v[idx] * np.log(v[a]))
        return E, {name: float(bar[i]) for i, name in self.params}

    def logWeights(self, defs: Dict[str, Any]) -> np.ndarray:
        w = self.values(defs)
        log = math.log
        return np.array([log(x) if x > 0 else (-math.inf if x == 0 else math.nan) for x in w.tolist()], np.float64)

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

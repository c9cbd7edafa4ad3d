"""Host-side machine model: the JSON transducer format, weight expressions, parameters.

This is *not* the product hot path; it is the minimum of the reference's L1/L2 layers
needed to turn a Machine Boss JSON machine + parameters into the numeric, flattened
``EvaluatedMachine`` that the HIP DP engine consumes.  It mirrors (does not copy):

* JSON machine reading          /root/reference/src/machine.cpp:446-506  (state / trans / to / in / out / weight / defs / cons)
* alphabets (sorted symbol sets) src/machine.cpp:175-191
* weight expression evaluation   src/weight.cpp:241-300, JSON opcodes src/weight.cpp:547-590
* default parameters             src/constraints.cpp:65-75 (norm group -> 1/n, prob -> 0.5, rate -> 1)
* getParamDefs(use-defaults)     src/machine.cpp:2022-2027
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional


class MachineError(RuntimeError):
    """Raised where the reference would Abort()/Fail() (src/util.cpp:39-57)."""


@dataclass
class MachineTransition:
    dest: int
    inp: str = ""      # "" == epsilon  (src/machine.h MachineTransition::in)
    out: str = ""      # "" == epsilon
    weight: Any = 1    # JSON weight expression

    def inputEmpty(self) -> bool:
        return self.inp == ""

    def outputEmpty(self) -> bool:
        return self.out == ""

    def isSilent(self) -> bool:
        return self.inp == "" and self.out == ""


@dataclass
class MachineState:
    name: Any = None
    trans: List[MachineTransition] = field(default_factory=list)

    def getTransition(self, ti: int) -> MachineTransition:
        return self.trans[ti]


@dataclass
class Constraints:
    prob: List[str] = field(default_factory=list)
    norm: List[List[str]] = field(default_factory=list)
    rate: List[str] = field(default_factory=list)

    @classmethod
    def fromJson(cls, j: dict) -> "Constraints":
        return cls(prob=list(j.get("prob", [])), norm=[list(g) for g in j.get("norm", [])],
                   rate=list(j.get("rate", [])))

    def defaultParams(self) -> Dict[str, Any]:
        """src/constraints.cpp:65-75."""
        p: Dict[str, Any] = {}
        for group in self.norm:
            for name in group:
                p[name] = 1.0 / float(len(group))
        for name in self.prob:
            p[name] = 0.5
        for name in self.rate:
            p[name] = 1
        return p

    def empty(self) -> bool:
        return not (self.prob or self.norm or self.rate)


def evalWeight(w: Any, defs: Dict[str, Any], _excluded: frozenset = frozenset()) -> float:
    """Numeric value of a JSON weight expression (src/weight.cpp:241-300).

    ``defs`` maps parameter names to numbers or to further expressions (function defs).
    Arithmetic is plain IEEE double in the operand order given, so that ``log(evalWeight(..))``
    reproduces the reference's transition log-weights bit for bit (needed for bit-exact Viterbi).
    """
    if w is None:
        return 0.0
    if isinstance(w, bool):
        return 1.0 if w else 0.0
    if isinstance(w, (int, float)):
        return float(w)
    if isinstance(w, str):
        if w not in defs or w in _excluded:
            raise MachineError("Parameter %s not defined" % w)
        v = defs[w]
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            return float(v)
        return evalWeight(v, defs, _excluded | {w})
    if isinstance(w, list):
        raise MachineError("Unexpected type in WeightExpr: array")
    if not isinstance(w, dict) or len(w) == 0:
        raise MachineError("WeightExpr must be JSON object with an opcode")
    op, args = next(iter(w.items()))
    ev = lambda x: evalWeight(x, defs, _excluded)
    if op == "log":
        return math.log(ev(args))
    if op == "exp":
        return math.exp(ev(args))
    if op == "not":
        return 1.0 - ev(args)
    if op == "geomsum":
        return 1.0 / (1.0 - ev(args))
    if op == "*":
        return ev(args[0]) * ev(args[1])
    if op == "/":
        return ev(args[0]) / ev(args[1])
    if op == "+":
        return ev(args[0]) + ev(args[1])
    if op == "-":
        return ev(args[0]) - ev(args[1])
    if op == "pow":
        return math.pow(ev(args[0]), ev(args[1]))
    raise MachineError("Unknown opcode %s in JSON" % op)


def weightParams(w: Any, defs: Dict[str, Any]) -> set:
    """Free parameter names appearing in ``w`` after expanding function defs (src/weight.cpp params())."""
    out: set = set()
    if isinstance(w, str):
        if w in defs and not isinstance(defs[w], (int, float)):
            out |= weightParams(defs[w], {k: v for k, v in defs.items() if k != w})
        else:
            out.add(w)
    elif isinstance(w, dict) and w:
        op, args = next(iter(w.items()))
        if isinstance(args, list):
            for a in args:
                out |= weightParams(a, defs)
        else:
            out |= weightParams(args, defs)
    return out


@dataclass
class Machine:
    state: List[MachineState] = field(default_factory=list)
    funcs: Dict[str, Any] = field(default_factory=dict)   # "defs"
    cons: Constraints = field(default_factory=Constraints)

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def fromJson(cls, pj: Any) -> "Machine":
        if isinstance(pj, str):
            pj = json.loads(pj)
        if "state" not in pj:
            raise MachineError("Only basic transducers (a 'state' array) are supported on this path; "
                               "compose/concat/... are machine-algebra operations outside the DP hot path")
        m = cls()
        if "defs" in pj:
            m.funcs = dict(pj["defs"])
        if "cons" in pj:
            m.cons = Constraints.fromJson(pj["cons"])
        jstate = pj["state"]
        id2n: Dict[str, int] = {}
        dup: set = set()
        for n, js in enumerate(jstate):
            if "n" in js and js["n"] != n:
                raise MachineError("StateIndex n=%d out of sequence" % js["n"])
            ms = MachineState()
            if "id" in js:
                sid = js["id"]
                if isinstance(sid, (int, float)) and not isinstance(sid, bool):
                    raise MachineError("id can't be a number")
                key = json.dumps(sid, sort_keys=False, separators=(",", ":"))
                if key in id2n:
                    dup.add(key)
                else:
                    id2n[key] = n
                ms.name = sid
            m.state.append(ms)
        for ms, js in zip(m.state, jstate):
            for jt in js.get("trans", []):
                to = jt["to"]
                if isinstance(to, int) and not isinstance(to, bool):
                    dest = to
                else:
                    key = json.dumps(to, sort_keys=False, separators=(",", ":"))
                    if key not in id2n:
                        raise MachineError("No such state in \"to\": %s" % key)
                    if key in dup:
                        raise MachineError("Ambiguous destination state ID in \"to\": %s" % key)
                    dest = id2n[key]
                if "weight" in jt:
                    wt = jt["weight"]
                elif "expr" in jt:
                    raise MachineError("string weight expressions ('expr') need the reference's PEG parser (out of scope)")
                else:
                    wt = 1
                ms.trans.append(MachineTransition(dest=dest, inp=jt.get("in", ""), out=jt.get("out", ""), weight=wt))
        for ms in m.state:
            for t in ms.trans:
                if not (0 <= t.dest < len(m.state)):
                    raise MachineError("State %d does not exist" % t.dest)
        return m

    @classmethod
    def fromFile(cls, path: str) -> "Machine":
        with open(path) as f:
            return cls.fromJson(json.load(f))

    # ---- queries ------------------------------------------------------------------------
    def nStates(self) -> int:
        return len(self.state)

    def nTransitions(self) -> int:
        return sum(len(s.trans) for s in self.state)

    def startState(self) -> int:
        return 0

    def endState(self) -> int:
        return len(self.state) - 1

    def inputAlphabet(self) -> List[str]:
        return sorted({t.inp for s in self.state for t in s.trans if t.inp != ""})

    def outputAlphabet(self) -> List[str]:
        return sorted({t.out for s in self.state for t in s.trans if t.out != ""})

    def isAdvancingMachine(self) -> bool:
        """src/machine.cpp:758-764 (note: the loop starts at state 1)."""
        for s in range(1, len(self.state)):
            for t in self.state[s].trans:
                if t.isSilent() and t.dest <= s:
                    return False
        return True

    def getParamDefs(self, assignDefaultValuesToMissingParams: bool = False) -> Dict[str, Any]:
        """src/machine.cpp:2022-2027: defaults from constraints, overwritten by the machine's own defs."""
        p: Dict[str, Any] = {}
        if assignDefaultValuesToMissingParams:
            p.update(self.cons.defaultParams())
        p.update(self.funcs)
        return p

    def stateNameJson(self, s: int) -> str:
        nm = self.state[s].name
        return str(s) if nm is None else json.dumps(nm, separators=(",", ":"))

"""HMMER3 profile -> generator machine (`boss --hmmer / --hmmer-global / --hmmer-plan7 / --hmmer-multihit`).

Host-side input format of the widening row SURVEY.md section 8(f)4 / BASELINE config 5: the profile is a one-tape
machine (no input), which the device library sweeps with its one-tape kernel family.  Follows (does not copy)
/root/reference/src/hmmer.h:12-56 (state numbering) and src/hmmer.cpp:10-251 (file parser, core machine in local /
global mode, Plan7 flanks, match occupancy); pinned by the reference goldens t/expect/fn3{,-plan7,-multihit}.json.
"""
from __future__ import annotations

import math
import re
from typing import List

from .machine import Machine, MachineError, MachineState, MachineTransition

# SwissProt background composition used for the N/C/J flanks (src/hmmer.cpp:22-41)
_BACKGROUND = dict(A=0.0825, C=0.0138, D=0.0546, E=0.0673, F=0.0386, G=0.0708, H=0.0227, I=0.0592, K=0.0581, L=0.0965,
                   M=0.0241, N=0.0405, P=0.0473, Q=0.0393, R=0.0553, S=0.0663, T=0.0535, V=0.0686, W=0.0109, Y=0.0292)


def _f32(x: float) -> float:
    import numpy as np
    return float(np.float32(x))


def strToProb(s: str) -> float:
    """src/hmmer.cpp:10-12: '*' is probability 0, anything else exp(-x) with x read through stof (single precision)."""
    return 0.0 if s == "*" else math.exp(-_f32(float(s)))


class HmmerNode:
    __slots__ = ("matchEmit", "insEmit", "m_to_m", "m_to_i", "m_to_d", "i_to_m", "i_to_i", "d_to_m", "d_to_d")


class HmmerModel:
    def __init__(self):
        self.node: List[HmmerNode] = []
        self.alph: List[str] = []
        self.ins0Emit: List[float] = []
        self.nullEmit: List[float] = []
        self.b_to_m1 = self.b_to_i0 = self.b_to_d1 = self.i0_to_m1 = self.i0_to_i0 = 0.0

    # ---- state numbering (src/hmmer.h:27-49) --------------------------------------------------------------------
    def b_idx(self): return 0
    def ix_idx(self, n): return 5 * n + 1
    def i_idx(self, n): return 5 * n + 2
    def mx_idx(self, n): return 5 * n - 2
    def m_idx(self, n): return 5 * n - 1
    def d_idx(self, n): return 5 * n
    def core_end_idx(self): return 5 * len(self.node) + 3
    def nCoreStates(self): return 5 * len(self.node) + 4
    def n_idx(self): return self.nCoreStates()
    def nx_idx(self): return self.nCoreStates() + 1
    def plan7_b_idx(self): return self.nCoreStates() + 2
    def cx_idx(self): return self.nCoreStates() + 3
    def c_idx(self): return self.nCoreStates() + 4
    def jx_idx(self): return self.nCoreStates() + 5
    def j_idx(self): return self.nCoreStates() + 6
    def t_idx(self): return self.nCoreStates() + 7
    def nPlan7States(self): return self.nCoreStates() + 8

    # ---- parser (format read by src/hmmer.cpp:43-105) -------------------------------------------------------------
    @classmethod
    def fromFile(cls, path: str) -> "HmmerModel":
        with open(path) as f:
            return cls.fromText(f.read())

    @classmethod
    def fromText(cls, text: str) -> "HmmerModel":
        """HMMER3 ASCII save file: everything up to the line starting with the tag `HMM` is header; that line lists the
        alphabet; two lines on (transition labels, optional COMPO) come the node-0 insert emissions and the begin
        transitions, then three lines per node (match emissions with the node number first and five annotation fields
        last, insert emissions, seven transitions) until `//`."""
        h = cls()
        rows = [ln.split() for ln in text.split("\n")]
        start = next((k for k, ln in enumerate(text.split("\n")) if re.match(r"^HMM(?![A-Z])", ln)), None)
        if start is None:
            h.loadNullModel()
            return h
        if len(rows[start]) <= 1:
            raise MachineError("HMM parse error: empty alphabet")
        h.alph = rows[start][1:]
        nSym = len(h.alph)
        body = rows[start + 3:]
        if not body:
            h.loadNullModel()
            return h
        if len(body[0]) != nSym:
            raise MachineError("HMM parse error: wrong number of fields in node 0 insert line")
        h.ins0Emit = [strToProb(f) for f in body[0]]
        if len(body) > 1:
            h.b_to_m1, h.b_to_i0, h.b_to_d1, h.i0_to_m1, h.i0_to_i0 = (strToProb(f) for f in body[1][:5])
        k = 2
        raw = text.split("\n")[start + 3:]
        while k < len(body) and not raw[k].startswith("//"):
            match = body[k]
            if len(match) != nSym + 6:
                raise MachineError("HMM parse error: wrong number of fields in node match line")
            if int(match[0]) != len(h.node) + 1:
                raise MachineError("HMM parse error: incorrect node index")
            if k + 2 >= len(body):
                raise MachineError("HMM parse error: premature truncation of node")
            ins, trans = body[k + 1], body[k + 2]
            if len(ins) != nSym:
                raise MachineError("HMM parse error: wrong number of fields in node insert line")
            if len(trans) != 7:
                raise MachineError("HMM parse error: wrong number of fields in node transitions line")
            nd = HmmerNode()
            nd.matchEmit = [strToProb(f) for f in match[1:nSym + 1]]
            nd.insEmit = [strToProb(f) for f in ins]
            for name, f in zip(HmmerNode.__slots__[2:], trans):
                setattr(nd, name, strToProb(f))
            h.node.append(nd)
            k += 3
        h.loadNullModel()
        return h

    def loadNullModel(self) -> None:
        self.nullEmit = [_BACKGROUND.get(sym, 1.0 / len(self.alph)) for sym in self.alph]

    def truncated(self, nNodes: int) -> "HmmerModel":
        """The first nNodes nodes of the profile ('use a truncated profile')."""
        h = HmmerModel()
        h.__dict__.update(self.__dict__)
        h.node = list(self.node[:nNodes])
        return h

    # ---- src/hmmer.cpp:237-251 ----------------------------------------------------------------------------------
    def calcMatchOccupancy(self) -> List[float]:
        nd = self.node
        mocc = [0.0] * len(nd)
        if len(nd) > 1:
            mocc[1] = nd[0].m_to_i + nd[0].m_to_m
        for k in range(2, len(nd)):
            mocc[k] = mocc[k - 1] * (nd[k].m_to_m + nd[k].m_to_i) + (1.0 - mocc[k - 1]) * nd[k].d_to_m
        return mocc

    # ---- core machine (same states, transition order and weights as src/hmmer.cpp:107-177) -----------------------------
    def machine(self, local: bool = True) -> Machine:
        """B, then per node k: Ix(k) I(k) | Mx(k) M(k) D(k), then E.  M(k) / I(k) emit and move to their `x` twin, which carries
        the node's outgoing transitions.  Local mode (p7_ProfileConfig): B enters M(k) with occupancy-weighted probability
        and every M(k), D(k) may leave to E with weight 1; global mode: B -> M1 / I0 / D1 and only the last node reaches E."""
        if not self.node:
            raise MachineError("Attempt to create a transducer from an empty HMMER model")
        N = len(self.node)
        E = self.core_end_idx()
        m = Machine()
        m.state = [MachineState() for _ in range(self.nCoreStates())]
        names = {self.b_idx(): "B", E: "E"}
        for k in range(N + 1):
            names[self.i_idx(k)] = "I%d" % k; names[self.ix_idx(k)] = "Ix%d" % k
            if k:
                names[self.m_idx(k)] = "M%d" % k; names[self.mx_idx(k)] = "Mx%d" % k; names[self.d_idx(k)] = "D%d" % k
        for idx, nm in names.items():
            m.state[idx].name = nm

        def arc(src, dst, w, out=""):
            m.state[src].trans.append(MachineTransition(dst, "", out, w))

        def emit(src, dst, probs):
            for sym, p in zip(self.alph, probs):
                arc(src, dst, p, sym)

        if local:
            occ = self.calcMatchOccupancy()
            Z = sum(occ[k] * (N - k + 1) for k in range(1, N))
            for k in range(1, N):
                arc(self.b_idx(), self.m_idx(k), occ[k] / Z)
        else:
            for dst, w in ((self.m_idx(1), self.b_to_m1), (self.i_idx(0), self.b_to_i0), (self.d_idx(1), self.b_to_d1)):
                arc(self.b_idx(), dst, w)
        arc(self.ix_idx(0), self.m_idx(1), self.i0_to_m1)
        arc(self.ix_idx(0), self.i_idx(0), self.i0_to_i0)
        emit(self.i_idx(0), self.ix_idx(0), self.ins0Emit)
        for k, nd in enumerate(self.node, 1):
            last = (k == N)
            nextM = E if last else self.m_idx(k + 1)
            # outgoing transitions of the node, from the post-emission states Mx / Ix and from D
            if not last or not local:
                arc(self.mx_idx(k), nextM, nd.m_to_m)
            arc(self.mx_idx(k), self.i_idx(k), nd.m_to_i)
            if not last:
                arc(self.mx_idx(k), self.d_idx(k + 1), nd.m_to_d)
            arc(self.ix_idx(k), nextM, nd.i_to_m)
            arc(self.ix_idx(k), self.i_idx(k), nd.i_to_i)
            if not last:
                arc(self.d_idx(k), nextM, nd.d_to_m)
                arc(self.d_idx(k), self.d_idx(k + 1), nd.d_to_d)
            elif not local:
                arc(self.d_idx(k), E, nd.d_to_m)
            # emissions, symbol by symbol: the match and the insert state alternate in the reference's loop, which only
            # fixes the order inside each state's own list
            emit(self.m_idx(k), self.mx_idx(k), nd.matchEmit)
            emit(self.i_idx(k), self.ix_idx(k), nd.insEmit)
            if local:
                arc(self.m_idx(k), E, 1)
                arc(self.d_idx(k), E, 1)
        return m

    # ---- Plan7 flanks (src/hmmer.cpp:179-235) -------------------------------------------------------------------
    def plan7Machine(self, multihit: bool = False, L: float = 400) -> Machine:
        if not self.node:
            raise MachineError("Attempt to create a Plan7 transducer from an empty HMMER model")
        core = self.machine(True)
        T = MachineTransition
        m = Machine()
        m.state = core.state + [MachineState() for _ in range(self.nPlan7States() - self.nCoreStates())]
        st = m.state
        st[self.plan7_b_idx()] = MachineState("B", list(st[self.b_idx()].trans))
        st[self.b_idx()] = MachineState("S", [T(self.nx_idx(), "", "", 1.0)])
        st[self.n_idx()].name = "N"
        for sym, p in zip(self.alph, self.nullEmit):
            st[self.n_idx()].trans.append(T(self.nx_idx(), "", sym, p))
        st[self.nx_idx()].name = "Nx"
        st[self.nx_idx()].trans.append(T(self.n_idx(), "", "", L / (L + 1)))
        st[self.nx_idx()].trans.append(T(self.plan7_b_idx(), "", "", 1.0 / (L + 1)))
        e = st[self.core_end_idx()]
        if multihit:
            e.trans.append(T(self.cx_idx(), "", "", 0.5))
            e.trans.append(T(self.jx_idx(), "", "", 0.5))
        else:
            e.trans.append(T(self.cx_idx(), "", "", 1.0))
        st[self.c_idx()].name = "C"
        for sym, p in zip(self.alph, self.nullEmit):
            st[self.c_idx()].trans.append(T(self.cx_idx(), "", sym, p))
        st[self.cx_idx()].name = "Cx"
        st[self.cx_idx()].trans.append(T(self.c_idx(), "", "", L / (L + 1)))
        st[self.cx_idx()].trans.append(T(self.t_idx(), "", "", 1.0 / (L + 1)))
        st[self.j_idx()].name = "J"
        st[self.jx_idx()].name = "Jx"
        if multihit:
            for sym, p in zip(self.alph, self.nullEmit):
                st[self.j_idx()].trans.append(T(self.jx_idx(), "", sym, p))
            st[self.jx_idx()].trans.append(T(self.j_idx(), "", "", L / (L + 1)))
            st[self.jx_idx()].trans.append(T(self.plan7_b_idx(), "", "", 1.0 / (L + 1)))
        st[self.t_idx()].name = "T"
        return m

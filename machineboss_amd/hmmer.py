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

    # ---- parser (src/hmmer.cpp:43-105) --------------------------------------------------------------------------
    @classmethod
    def fromFile(cls, path: str) -> "HmmerModel":
        with open(path) as f:
            return cls.fromText(f.read())

    @classmethod
    def fromText(cls, text: str) -> "HmmerModel":
        h = cls()
        lines = text.split("\n")
        k = 0
        while k < len(lines):
            line = lines[k]; k += 1
            tag = re.match(r"^([A-Z]+)", line)
            if not tag or tag.group(1) != "HMM":
                continue
            fields = line.split()
            if len(fields) <= 1:
                raise MachineError("HMM parse error: empty alphabet")
            h.alph = fields[1:]
            k += 2                              # the transition-label line and the optional COMPO line
            if k >= len(lines): break
            ins0 = lines[k].split(); k += 1
            if len(ins0) != len(h.alph):
                raise MachineError("HMM parse error: wrong number of fields in node 0 insert line")
            h.ins0Emit = [strToProb(s) for s in ins0]
            if k >= len(lines): break
            bt = lines[k].split(); k += 1
            h.b_to_m1, h.b_to_i0, h.b_to_d1, h.i0_to_m1, h.i0_to_i0 = (strToProb(s) for s in bt[:5])
            while k < len(lines):
                line = lines[k]; k += 1
                if line.startswith("//"):
                    break
                ml = line.split()
                if len(ml) != len(h.alph) + 6:
                    raise MachineError("HMM parse error: wrong number of fields in node match line")
                if int(ml[0]) != len(h.node) + 1:
                    raise MachineError("HMM parse error: incorrect node index")
                if k + 1 >= len(lines):
                    raise MachineError("HMM parse error: premature truncation of node")
                il = lines[k].split(); tl = lines[k + 1].split(); k += 2
                if len(il) != len(h.alph):
                    raise MachineError("HMM parse error: wrong number of fields in node insert line")
                if len(tl) != 7:
                    raise MachineError("HMM parse error: wrong number of fields in node transitions line")
                n = HmmerNode()
                n.matchEmit = [strToProb(s) for s in ml[1:len(h.alph) + 1]]
                n.insEmit = [strToProb(s) for s in il]
                n.m_to_m, n.m_to_i, n.m_to_d, n.i_to_m, n.i_to_i, n.d_to_m, n.d_to_d = (strToProb(s) for s in tl)
                h.node.append(n)
            break
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

    # ---- core machine (src/hmmer.cpp:107-177) -------------------------------------------------------------------
    def machine(self, local: bool = True) -> Machine:
        if not self.node:
            raise MachineError("Attempt to create a transducer from an empty HMMER model")
        N = len(self.node)
        m = Machine()
        m.state = [MachineState() for _ in range(self.nCoreStates())]
        T = MachineTransition
        st = m.state
        st[self.b_idx()].name = "B"
        if local:
            occ = self.calcMatchOccupancy()
            Z = 0.0
            for k in range(1, N):
                Z += occ[k] * (N - k + 1)
            for k in range(1, N):
                st[self.b_idx()].trans.append(T(self.m_idx(k), "", "", occ[k] / Z))
        else:
            st[self.b_idx()].trans.append(T(self.m_idx(1), "", "", self.b_to_m1))
            st[self.b_idx()].trans.append(T(self.i_idx(0), "", "", self.b_to_i0))
            st[self.b_idx()].trans.append(T(self.d_idx(1), "", "", self.b_to_d1))
        st[self.ix_idx(0)].trans.append(T(self.m_idx(1), "", "", self.i0_to_m1))
        st[self.ix_idx(0)].trans.append(T(self.i_idx(0), "", "", self.i0_to_i0))
        for sym, p in zip(self.alph, self.ins0Emit):
            st[self.i_idx(0)].trans.append(T(self.ix_idx(0), "", sym, p))
        for n in range(N + 1):
            st[self.i_idx(n)].name = "I%d" % n
            st[self.ix_idx(n)].name = "Ix%d" % n
            if n == 0:
                continue
            nd = self.node[n - 1]
            st[self.m_idx(n)].name = "M%d" % n
            st[self.mx_idx(n)].name = "Mx%d" % n
            st[self.d_idx(n)].name = "D%d" % n
            end = (n == N)
            mx, ix, d = st[self.mx_idx(n)], st[self.ix_idx(n)], st[self.d_idx(n)]
            if end:
                if not local:
                    mx.trans.append(T(self.core_end_idx(), "", "", nd.m_to_m))
            else:
                mx.trans.append(T(self.m_idx(n + 1), "", "", nd.m_to_m))
            mx.trans.append(T(self.i_idx(n), "", "", nd.m_to_i))
            if not end:
                mx.trans.append(T(self.d_idx(n + 1), "", "", nd.m_to_d))
            ix.trans.append(T(self.core_end_idx() if end else self.m_idx(n + 1), "", "", nd.i_to_m))
            ix.trans.append(T(self.i_idx(n), "", "", nd.i_to_i))
            if end:
                if not local:
                    d.trans.append(T(self.core_end_idx(), "", "", nd.d_to_m))
            else:
                d.trans.append(T(self.m_idx(n + 1), "", "", nd.d_to_m))
                d.trans.append(T(self.d_idx(n + 1), "", "", nd.d_to_d))
            for sym, pm, pi in zip(self.alph, nd.matchEmit, nd.insEmit):
                st[self.m_idx(n)].trans.append(T(self.mx_idx(n), "", sym, pm))
                st[self.i_idx(n)].trans.append(T(self.ix_idx(n), "", sym, pi))
            if local:
                st[self.m_idx(n)].trans.append(T(self.core_end_idx(), "", "", 1))
                d.trans.append(T(self.core_end_idx(), "", "", 1))
        st[self.core_end_idx()].name = "E"
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

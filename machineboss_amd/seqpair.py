"""SeqPair, SeqPairList and Envelope -- the data formats either side of the DP path.

Mirrors /root/reference/src/seqpair.{h,cpp}: a SeqPair is two named symbol sequences plus an optional alignment
(a list of [inputSymbol, outputSymbol] columns, "" = gap); an Envelope is, per output position y, the half-open
interval [inStart[y], inEnd[y]) of input positions whose cells exist (src/seqpair.h:75-97).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

from .machine import MachineError


@dataclass
class SeqPair:
    """src/seqpair.h:56-73."""
    input: List[str]
    output: List[str]
    inputName: str = "input"
    outputName: str = "output"
    alignment: List[Tuple[str, str]] = field(default_factory=list)
    metadata: Any = None

    @classmethod
    def fromJson(cls, j: dict) -> "SeqPair":
        """SeqPair::readJson (src/seqpair.cpp:8-38): with an alignment, missing sequences default to its columns."""
        def named(key, default_seq):
            nj = j.get(key)
            if nj is None:
                return key, list(default_seq)
            name = nj.get("name", key)
            seq = nj.get("sequence")
            if seq is None:
                if default_seq is None:
                    raise MachineError("Sequence %s has no symbols" % key)
                seq = default_seq
            return name, list(seq)
        if "alignment" in j:
            ali = [(str(c[0]), str(c[1])) for c in j["alignment"]]
            ins = [a for a, _ in ali if a]; outs = [b for _, b in ali if b]
            iname, iseq = named("input", ins); oname, oseq = named("output", outs)
            return cls(iseq, oseq, iname, oname, ali, j.get("meta"))
        iname, iseq = named("input", None); oname, oseq = named("output", None)
        return cls(iseq, oseq, iname, oname)

    def toJson(self) -> dict:
        j = {"input": {"name": self.inputName, "sequence": list(self.input)},
             "output": {"name": self.outputName, "sequence": list(self.output)}}
        if self.alignment:
            j["alignment"] = [[a, b] for a, b in self.alignment]
        if self.metadata is not None:
            j["meta"] = self.metadata
        return j


def seqPairListFromJson(j: Sequence[dict]) -> List[SeqPair]:
    """SeqPairList::readJson (src/seqpair.cpp:244-249)."""
    return [SeqPair.fromJson(x) for x in j]


class Envelope:
    """src/seqpair.h:75-121, src/seqpair.cpp:100-229."""

    def __init__(self, sp: Optional[SeqPair] = None, width: Optional[int] = None):
        self.clear()
        if sp is not None:
            if sp.alignment:
                if width is None:
                    self.initPath(sp.alignment)
                else:
                    self.initPathArea(sp.alignment, width)
            else:
                self.initFull(sp)
            if not self.fits(sp):
                raise MachineError("Envelope/sequence mismatch")

    def clear(self):
        self.inLen = self.outLen = 0
        self.inStart: List[int] = [0]
        self.inEnd: List[int] = [1]

    def initFull(self, sp: SeqPair):
        self.clear()
        self.inLen, self.outLen = len(sp.input), len(sp.output)
        self.inStart = [0] * (self.outLen + 1)
        self.inEnd = [self.inLen + 1] * (self.outLen + 1)

    def initPath(self, cols: Sequence[Tuple[str, str]]):
        """src/seqpair.cpp:134-152: exactly the cells the alignment path visits."""
        self.clear()
        for a, b in cols:
            gotIn, gotOut = bool(a), bool(b)
            if not gotIn and gotOut:
                self.inStart.append(self.inEnd[-1] - 1); self.inEnd.append(self.inEnd[-1]); self.outLen += 1
            elif gotIn and not gotOut:
                self.inEnd[-1] += 1; self.inLen += 1
            elif gotIn and gotOut:
                self.inStart.append(self.inEnd[-1]); self.inEnd.append(self.inEnd[-1] + 1)
                self.inLen += 1; self.outLen += 1

    def initPathArea(self, cols: Sequence[Tuple[str, str]], width: int):
        """src/seqpair.cpp:154-182: everything within `width` matches of the alignment."""
        self.clear()
        match: List[int] = []; nBefore: List[int] = [0]
        for a, b in cols:
            gotIn, gotOut = bool(a), bool(b)
            if gotIn and gotOut:
                match.append(self.inLen)
            if gotIn:
                self.inLen += 1
            if gotOut:
                self.outLen += 1
                nBefore.append(len(match))
        self.inStart, self.inEnd = [], []
        for j in range(self.outLen + 1):
            iStart, iEnd = 0, self.inLen + 1
            if nBefore[j] > width:
                iStart = match[nBefore[j] - width - 1] + 1
            nAfter = len(match) - nBefore[j]
            if nAfter > width:
                iEnd = match[nBefore[j] + width] + 1
            self.inStart.append(iStart); self.inEnd.append(iEnd)

    def contains(self, x: int, y: int) -> bool:
        return 0 <= y <= self.outLen and self.inStart[y] <= x < self.inEnd[y]

    @staticmethod
    def overlapping(s1: int, e1: int, s2: int, e2: int) -> bool:
        return not (s1 >= e2 or s2 >= e1)

    def fits(self, sp: SeqPair) -> bool:
        return self.inLen == len(sp.input) and self.outLen == len(sp.output)

    def connected(self) -> bool:
        conn = self.overlapping(self.inStart[0], self.inEnd[0], 0, 1)
        for y in range(1, self.outLen + 1):
            conn = conn and self.overlapping(self.inStart[y - 1], self.inEnd[y - 1] + 1, self.inStart[y], self.inEnd[y])
        return conn and self.overlapping(self.inStart[self.outLen], self.inEnd[self.outLen], self.inLen, self.inLen + 1)

    def offsets(self) -> List[int]:
        """offsets[y] = number of supercells in rows < y (src/seqpair.cpp:195-204): the compact storage index base."""
        out = [0]
        for y in range(self.outLen + 1):
            out.append(out[-1] + self.inEnd[y] - self.inStart[y])
        return out

    def isFull(self) -> bool:
        return all(s == 0 for s in self.inStart) and all(e == self.inLen + 1 for e in self.inEnd)

    def writeJson(self) -> str:
        return "[" + ",".join("[%d,%d]" % (s, e) for s, e in zip(self.inStart, self.inEnd)) + "]"

    @classmethod
    def fullEnvelope(cls, sp: SeqPair) -> "Envelope":
        e = cls(); e.initFull(sp); return e

    @classmethod
    def pathEnvelope(cls, path: Sequence[Tuple[str, str]]) -> "Envelope":
        e = cls(); e.initPath(path); return e

    @classmethod
    def pathAreaEnvelope(cls, path: Sequence[Tuple[str, str]], width: int) -> "Envelope":
        e = cls(); e.initPathArea(path, width); return e

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
from .seqpair import Envelope, SeqPair


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


class Mt19937:
    """std::mt19937(seed) as the reference uses it (``generator()`` -> 32-bit outputs; src/util.h:102-106).

    ``result_bits`` is the width of ``std::mt19937::result_type`` (``uint_fast32_t``) on the platform the reference is built
    on.  random_double divides by ``numeric_limits<result_type>::max() + 1``, NOT by ``Generator::max() + 1``: with
    libstdc++ on LP64 Linux -- this container and the GPU box -- uint_fast32_t is a 64-bit unsigned long, so the variate lies
    in [0, 2**-32) and random_index all but always returns the first candidate of non-negligible weight.  That is the
    reference's behaviour here, hence the default (quirk Q12); ``result_bits=32`` is the libc++ (macOS) reading,
    under which samplePath really samples."""

    def __init__(self, seed: int = 5489, result_bits: int = 64):
        self._rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        self.result_type_max = (1 << result_bits) - 1

    def __call__(self) -> int:
        return int(self._rs.randint(0, 2 ** 32, dtype=np.uint64))

    @staticmethod
    def max() -> int:
        return 0xFFFFFFFF


def random_double(rng) -> float:
    """src/util.h:102-106: generator() / (numeric_limits<result_type>::max() + 1)."""
    return rng() / (float(getattr(rng, "result_type_max", rng.max())) + 1.0)


def random_index(weights: Sequence[float], rng) -> int:
    """src/util.h:151-165."""
    norm = 0.0
    for w in weights:
        if not w >= 0:
            raise MachineError("Negative weights in random_index")
        norm += w
    if not norm > 0:
        raise MachineError("Zero weights in random_index")
    variate = random_double(rng) * norm
    for n, w in enumerate(weights):
        variate -= w
        if variate <= 0:
            return n
    return len(weights)


def selectMaxTrans(logWeights: Sequence[float]) -> int:
    """DPMatrix::selectMaxTrans (src/dpmatrix.defs.h:171-174): index of the FIRST maximum (std::max_element)."""
    best, bi = None, 0
    for k, w in enumerate(logWeights):
        if best is None or w > best:
            best, bi = w, k
    return bi


def randomTransSelector(rng):
    """DPMatrix::randomTransSelector (src/dpmatrix.defs.h:176-186)."""
    return lambda logWeights: random_index([math.exp(lw) for lw in logWeights], rng)


class _TransMaps:
    """The reference's per-state ``incoming`` / ``outgoing`` maps (src/eval.h:59-70) rebuilt on the host from the flat
    edge arrays: key (state, inTok, outTok) -> [(otherState, transIndex, logWeight)] in multimap order
    (other state ascending, then insertion order = ascending transIndex, src/eval.cpp:60-61)."""

    def __init__(self, em: EvaluatedMachine):
        self.incoming: Dict[tuple, list] = {}
        self.outgoing: Dict[tuple, list] = {}
        for e in em.incomingOrder():
            e = int(e)
            self.incoming.setdefault((int(em.dst[e]), int(em.inTok[e]), int(em.outTok[e])), []).append(
                (int(em.src[e]), int(em.transIndex[e]), float(em.logWeight[e])))
        for e in em.outgoingOrder():
            e = int(e)
            self.outgoing.setdefault((int(em.src[e]), int(em.inTok[e]), int(em.outTok[e])), []).append(
                (int(em.dst[e]), int(em.transIndex[e]), float(em.logWeight[e])))


def _transMaps(em: EvaluatedMachine) -> _TransMaps:
    tm = getattr(em, "_transMaps", None)
    if tm is None:
        tm = _TransMaps(em)
        em._transMaps = tm
    return tm


class _DPMatrix:
    """DPMatrix<IdentityIndexMapper> (src/dpmatrix.h:64-163): full matrix, cell() = -inf outside the lattice.

    The fill runs on the GPU (mb_fill); the path functions below (traceBack / traceForward and their selectors and
    terminators, src/dpmatrix.defs.h:61-186) walk the finished matrix on the host exactly as the reference does --
    they visit O(path length x in-degree) cells and are only used by Machine::downsample / stochasticDownsample and
    ForwardMatrix::samplePath.  The Viterbi path of a batch (the hot use) is traced on the device instead.

    The fp64 matrix is fetched LAZILY, by the first cell() / cells() / writeJson / walker that reads it (``matrix_fills``
    counts the fetches): what the reference's callers read from a ViterbiMatrix or a ForwardMatrix -- logLike(), path(m)
    (target/boss.cpp:826-833, src/api.cpp:31-66) -- comes from the matrix-free sweeps the constructors run (mb_dp.hpp, the
    C++ twin of this module, does the same)."""
    _mode = capi.MB_FORWARD
    matrix_fills = 0          # class-wide count of fp64 matrices fetched from the device

    def __init__(self, machine: EvaluatedMachine, seqPair: SeqPair, envelope: Optional[Envelope] = None, startState: int = 0):
        self.machine, self.seqPair = machine, seqPair
        self.input = machine.inputTokenizer.tokenize(seqPair.input)     # raises like Tokenizer::tokenize
        self.output = machine.outputTokenizer.tokenize(seqPair.output)
        self.inLen, self.outLen, self.nStates = len(self.input), len(self.output), machine.nStates
        # Quirk Q1: the reference's constructors initialise their IndexMapper from the SeqPair, not from the Envelope
        # argument (src/dpmatrix.defs.h:16-17): the envelope is ALWAYS Envelope(seqPair) -- the alignment's path
        # envelope when the pair carries one, else full (src/seqpair.cpp:104-110).  `envelope` is accepted and ignored.
        self.env = Envelope(seqPair)
        if not self.env.connected():                                     # DPMatrix::alloc, src/dpmatrix.defs.h:31-32
            raise MachineError("Envelope is not connected:\n%s\n" % self.env.writeJson())
        if not self.env.fits(seqPair):
            raise MachineError("Envelope/sequence mismatch")
        self._dm = _device_machine(machine)
        self._startState = startState
        self._matrix = None

    @property
    def _cells(self) -> np.ndarray:
        if self._matrix is None:
            full = self.env.isFull()
            self._matrix = self._dm.fill(self._mode, self.input, self.output, self._startState,
                                         None if full else self.env.inStart, None if full else self.env.inEnd)  # [o][i][s]
            _DPMatrix.matrix_fills += 1
        return self._matrix

    def matrixFetched(self) -> bool:
        return self._matrix is not None

    def _onePairBatch(self) -> "capi.DeviceBatch":
        b = capi.DeviceBatch.from_pairs(self._dm, [(self.input, self.output)])
        if not self.env.isFull():      # Envelope(seqPair), quirk Q1
            b.set_envelopes([(self.env.inStart, self.env.inEnd)])
        return b

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

    # ---- DPMatrix::writeJson (src/dpmatrix.defs.h:39-53): every cell at setprecision(5), input position outermost ----
    def writeJson(self) -> str:
        import json as _json
        rows = []
        for i in range(self.inLen + 1):
            for o in range(self.outLen + 1):
                for s in range(self.nStates):
                    v = self.cell(i, o, s)
                    txt = ("%.5g" % v) if math.isfinite(v) else ("-inf" if v < 0 else "inf")
                    rows.append('  { "inPos": %d, "outPos": %d, "state": %s, "logLike": %s }'
                                % (i, o, _json.dumps(self.machine.stateNames[s], separators=(",", ":")), txt))
        return ('{\n "input": "%s",\n "output": "%s",\n "cell": [\n' % (self.seqPair.inputName, self.seqPair.outputName)
                + ",\n".join(rows) + "\n ]\n}\n")

    # ---- DPMatrix::pathIterate (src/dpmatrix.h:111-119): candidates of one label group, in multimap order ----------
    def _pathIterate(self, tmap: Dict[tuple, list], state: int, inTok: int, outTok: int, inPos: int, outPos: int,
                     states: list, transIndex: list, loglike: list):
        for other, ti, lw in tmap.get((state, inTok, outTok), ()):
            states.append(other); transIndex.append(ti); loglike.append(self.cell(inPos, outPos, other) + lw)

    # ---- traceBack (src/dpmatrix.defs.h:61-110) ------------------------------------------------------------------
    def traceBackFrom(self, m: Machine, inPos: int, outPos: int, s: int, stopTrace, selectTrans=selectMaxTrans) -> None:
        """The TraceTerminator overload: honours its position arguments; stopTrace(inPos,outPos,src,transIndex)->bool."""
        if not (self.cell(inPos, outPos, s) > -math.inf):
            raise MachineError("Can't do traceback: no finite-weight paths")
        inc = _transMaps(self.machine).incoming
        while inPos > 0 or outPos > 0 or s != 0:
            loglike: list = []; source: list = []; tidx: list = []
            inTok = int(self.input[inPos - 1]) if inPos else 0
            outTok = int(self.output[outPos - 1]) if outPos else 0
            if inPos and outPos:
                self._pathIterate(inc, s, inTok, outTok, inPos - 1, outPos - 1, source, tidx, loglike)
            if inPos:
                self._pathIterate(inc, s, inTok, 0, inPos - 1, outPos, source, tidx, loglike)
            if outPos:
                self._pathIterate(inc, s, 0, outTok, inPos, outPos - 1, source, tidx, loglike)
            self._pathIterate(inc, s, 0, 0, inPos, outPos, source, tidx, loglike)
            if not loglike:   # the reference would index an empty vector here (undefined behaviour)
                raise MachineError("Traceback reached a cell without incoming transitions")
            best = selectTrans(loglike)
            bestSource, bestTi = source[best], tidx[best]
            bestTrans = m.state[bestSource].getTransition(bestTi)
            if bestTrans.inp:
                inPos -= 1
            if bestTrans.out:
                outPos -= 1
            s = bestSource
            if stopTrace(inPos, outPos, s, bestTi):
                break

    def traceBack(self, m: Machine, s: Optional[int] = None, selectTrans=selectMaxTrans) -> MachinePath:
        """The MachinePath overloads.  Like the reference's 5-argument overload they always start at
        (inLen, outLen) -- src/dpmatrix.defs.h:72-80 ignores its position arguments."""
        path = MachinePath()

        def stop(ip, op, src, ti):
            path.trans.insert(0, m.state[src].getTransition(ti)); path.steps.insert(0, (src, ti))
            return False
        self.traceBackFrom(m, self.inLen, self.outLen, self.nStates - 1 if s is None else s, stop, selectTrans)
        return path

    # ---- traceForward (src/dpmatrix.defs.h:112-159) --------------------------------------------------------------
    def traceForwardFrom(self, m: Machine, inPos: int, outPos: int, s: int, stopTrace, selectTrans=selectMaxTrans) -> None:
        if not (self.cell(inPos, outPos, s) > -math.inf):
            raise MachineError("Can't do traceforward: no finite-weight paths")
        outg = _transMaps(self.machine).outgoing
        while inPos < self.inLen or outPos < self.outLen or s != self.nStates - 1:
            loglike: list = []; dest: list = []; tidx: list = []
            endIn, endOut = inPos == self.inLen, outPos == self.outLen
            inTok = 0 if endIn else int(self.input[inPos])
            outTok = 0 if endOut else int(self.output[outPos])
            if not endIn and not endOut:
                self._pathIterate(outg, s, inTok, outTok, inPos + 1, outPos + 1, dest, tidx, loglike)
            if not endIn:
                self._pathIterate(outg, s, inTok, 0, inPos + 1, outPos, dest, tidx, loglike)
            if not endOut:
                self._pathIterate(outg, s, 0, outTok, inPos, outPos + 1, dest, tidx, loglike)
            self._pathIterate(outg, s, 0, 0, inPos, outPos, dest, tidx, loglike)
            if not loglike:
                raise MachineError("Traceforward reached a cell without outgoing transitions")
            best = selectTrans(loglike)
            bestDest, bestTi = dest[best], tidx[best]
            if stopTrace(inPos, outPos, s, bestTi):
                break
            bestTrans = m.state[s].getTransition(bestTi)
            if bestTrans.dest != bestDest:
                raise MachineError("Traceforward error")
            if bestTrans.inp:
                inPos += 1
            if bestTrans.out:
                outPos += 1
            s = bestDest

    def traceForward(self, m: Machine, inPos: Optional[int] = None, outPos: int = 0, s: int = 0,
                     selectTrans=selectMaxTrans) -> MachinePath:
        """MachinePath overloads, quirks kept: without a position this is ``traceBack(m,0,0,0,...)`` --
        a traceback from state 0 (src/dpmatrix.defs.h:112-115); with one, the walk still starts at (inLen, outLen)
        (:118-126)."""
        if inPos is None:
            return self.traceBack(m, 0, selectTrans)
        path = MachinePath()

        def stop(ip, op, src, ti):
            path.trans.append(m.state[src].getTransition(ti)); path.steps.append((src, ti))
            return False
        self.traceForwardFrom(m, self.inLen, self.outLen, s, stop, selectTrans)
        return path


class ForwardMatrix(_DPMatrix):
    """src/forward.h:19-27.  The constructor runs the rolling sweep (log-likelihood only); the matrix comes on demand."""
    _mode = capi.MB_FORWARD

    def __init__(self, machine: EvaluatedMachine, seqPair: SeqPair, envelope: Optional[Envelope] = None, startState: int = 0):
        super().__init__(machine, seqPair, envelope, startState)
        self._ll = None
        self._rollable = startState == 0            # a caller-chosen start state (src/forward.h:24) exists on the matrix route only

    def logLike(self) -> float:
        """The matrix's end cell once the matrix has been fetched (posteriors normalised with it agree with the cells); before that the
        rolling sweep's value, computed on the first call -- nothing is swept at construction."""
        if self._matrix is not None or not self._rollable:
            return self.endCell()
        if self._ll is None:
            b = self._onePairBatch()
            self._ll = float(b.forward(capi.MB_ROLLING)[0])
            b.close()
        return self._ll

    def samplePath(self, m: Machine, rng, s: Optional[int] = None) -> MachinePath:
        """Stochastic traceback with exp(candidate) weights (src/forward.cpp:17-23)."""
        return self.traceBack(m, s, randomTransSelector(rng))


@dataclass(order=True)
class PostTrans:
    """BackwardMatrix::PostTrans (src/backward.h:20-27): ordered by posterior weight."""
    weight: float
    inPos: int = field(compare=False)
    outPos: int = field(compare=False)
    src: int = field(compare=False)
    transIndex: int = field(compare=False)


class BackwardMatrix(_DPMatrix):
    """src/backward.h:44-59.  MachineCounts runs the count sweep on the device for whole batches; the visitor forms
    here (getCounts with a callback, postTransQueue, traceFrom) serve Machine::downsample and walk the two host
    copies of the matrices like the reference (src/backward.cpp:52-108)."""
    _mode = capi.MB_BACKWARD

    def logLike(self) -> float:
        return self.startCell()

    def getCounts(self, forward: "ForwardMatrix", visit) -> None:
        """visit(src, transIndex, inPos, outPos, postProb) for every cell and outgoing edge, in the reference's order
        (src/backward.cpp:62-87); ``visit`` may also be a MachineCounts, whose flat count vector is then updated."""
        if isinstance(visit, MachineCounts):
            mc = visit
            off = self.machine.transOffset

            def visit(s, ti, ip, op, pp):   # BackwardMatrix::transitionCounter
                mc._flat[off[s] + ti] += pp
        outg = _transMaps(self.machine).outgoing
        ll = self.logLike()
        for outPos in range(self.outLen, -1, -1):
            endOut = outPos == self.outLen
            outTok = 0 if endOut else int(self.output[outPos])
            for inPos in range(self.env.inEnd[outPos] - 1, self.env.inStart[outPos] - 1, -1):
                endIn = inPos == self.inLen
                inTok = 0 if endIn else int(self.input[inPos])
                for s in range(self.nStates - 1, -1, -1):
                    logOdds = forward.cell(inPos, outPos, s) - ll
                    groups = []
                    if not endIn and not endOut:
                        groups.append((inTok, outTok, inPos + 1, outPos + 1))
                    if not endIn:
                        groups.append((inTok, 0, inPos + 1, outPos))
                    if not endOut:
                        groups.append((0, outTok, inPos, outPos + 1))
                    groups.append((0, 0, inPos, outPos))
                    for it, ot, ip, op in groups:
                        for dest, ti, lw in outg.get((s, it, ot), ()):
                            visit(s, ti, ip, op, math.exp(logOdds + (self.cell(ip, op, dest) + lw)))   # iterate forms cell + logWeight first (src/dpmatrix.h:111)

    def postTransQueue(self, forward: "ForwardMatrix") -> List[PostTrans]:
        """All posterior transition usages as a max-heap ordered list, largest weight first (src/backward.cpp:52-56).
        Returned sorted descending; ``pop(0)`` is priority_queue::top()+pop()."""
        q: List[PostTrans] = []
        self.getCounts(forward, lambda s, ti, ip, op, pp: q.append(PostTrans(pp, ip, op, s, ti)))
        q.sort(key=lambda pt: -pt.weight)
        return q

    def traceFrom(self, m: Machine, forward: "ForwardMatrix", inPos: int, outPos: int, state: int,
                  transIndex: Optional[int] = None, stopTrace=None):
        """src/backward.cpp:89-108.  With ``stopTrace`` it is the terminator overload (returns None); otherwise the
        concatenated MachinePath through (state[, transIndex])."""
        if stopTrace is not None:
            if not stopTrace(inPos, outPos, state, transIndex):
                forward.traceBackFrom(m, inPos, outPos, state, stopTrace)
                mt = m.state[state].getTransition(transIndex)
                self.traceForwardFrom(m, inPos + (1 if mt.inp else 0), outPos + (1 if mt.out else 0), mt.dest, stopTrace)
            return None
        path = forward.traceBack(m, state)            # MachinePath overloads: quirk Q2, see traceBack
        if transIndex is not None:
            path.trans.append(m.state[state].getTransition(transIndex)); path.steps.append((state, transIndex))
        fwd = self.traceForward(m, inPos, outPos, state)
        path.trans += fwd.trans; path.steps += fwd.steps
        return path


class ViterbiMatrix(_DPMatrix):
    """src/viterbi.h:9-18."""
    _mode = capi.MB_VITERBI

    def __init__(self, machine: EvaluatedMachine, seqPair: SeqPair, envelope: Optional[Envelope] = None):
        super().__init__(machine, seqPair, envelope, 0)
        # score + the fill's own arg-max chain through the family's traceback-byte / traceback-code sweep: no fp64 matrix
        b = self._onePairBatch()
        ll, off, edges = b.viterbi(paths=True)
        b.close()
        self._score = float(ll[0])
        self._edges = np.array(edges[off[0]:off[1]], copy=True)

    def logLike(self) -> float:
        return self._score          # = endCell(), bit for bit (the max semiring is exact in every kernel family)

    def path(self, m: Machine) -> MachinePath:
        """traceBack(m) (src/viterbi.cpp:49-51, dpmatrix.defs.h:61-110), traced on the device by the constructor."""
        if not (self._score > -math.inf):
            raise MachineError("Can't do traceback: no finite-weight paths")
        return edgesToPath(self.machine, m, self._edges)


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
        envs = [Envelope(sp) for sp, k in zip(pairs, ok) if k]      # quirk Q1: RollingOutputForwardMatrix(eval, seqPair) uses Envelope(seqPair)
        if any(not e.isFull() for e in envs):
            b.set_envelopes([None if e.isFull() else (e.inStart, e.inEnd) for e in envs])
        ll = b.forward(capi.MB_ROLLING if rolling else capi.MB_MATERIALISE)
        b.close()
        it = iter(ll)
        for k, good in enumerate(ok):
            if good:
                out[k] = float(next(it))
    return out


def viterbiBatch(machine: EvaluatedMachine, m: Machine, pairs: Sequence[SeqPair]) -> List[tuple]:
    """The `--viterbi/--align` loop of target/boss.cpp:822-848 as one device call: per pair (logLike, MachinePath or
    None); pairs that cannot be tokenised score -inf, and a -inf score has no path."""
    dm = _device_machine(machine)
    ok = [machine.canTokenize(sp.input, sp.output) for sp in pairs]
    toks = [(machine.inputTokenizer.tokenize(sp.input), machine.outputTokenizer.tokenize(sp.output))
            for sp, k in zip(pairs, ok) if k]
    out: List[tuple] = [(-math.inf, None)] * len(pairs)
    if toks:
        b = capi.DeviceBatch.from_pairs(dm, toks)
        envs = [Envelope(sp) for sp, k in zip(pairs, ok) if k]      # quirk Q1: aligned pairs use their path envelope
        if any(not e.isFull() for e in envs):
            b.set_envelopes([None if e.isFull() else (e.inStart, e.inEnd) for e in envs])
        ll, off, edges = b.viterbi(paths=True)
        b.close()
        j = 0
        for k, good in enumerate(ok):
            if good:
                v = float(ll[j])
                out[k] = (v, edgesToPath(machine, m, edges[off[j]:off[j + 1]]) if v > -math.inf else None)
                j += 1
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

    @staticmethod
    def deviceBatch(machine: EvaluatedMachine, seqPairs: Sequence[SeqPair]) -> "capi.DeviceBatch":
        """Tokenise a SeqPairList once and park it in HBM (with the envelopes of its aligned pairs): Baum-Welch reuses
        the same batch in every iteration, only the machine's weights change."""
        dm = _device_machine(machine)
        toks = [(machine.inputTokenizer.tokenize(sp.input), machine.outputTokenizer.tokenize(sp.output)) for sp in seqPairs]
        b = capi.DeviceBatch.from_pairs(dm, toks)
        envs = [Envelope(sp) for sp in seqPairs]      # quirk Q1: always Envelope(seqPair), whatever the caller passes
        for e in envs:
            if not e.connected():
                raise MachineError("Envelope is not connected:\n%s\n" % e.writeJson())
        if any(not e.isFull() for e in envs):
            b.set_envelopes([None if e.isFull() else (e.inStart, e.inEnd) for e in envs])
        return b

    def addDeviceBatch(self, b: "capi.DeviceBatch") -> List[float]:
        _, s, ll = b.counts(self._flat)
        self.loglike += s
        return [float(x) for x in ll]

    def addBatch(self, seqPairs: Sequence[SeqPair], envelopes: Optional[Sequence[Envelope]] = None) -> List[float]:
        """MachineCounts::add over a list (src/counts.cpp:37-64).  Quirk Q1: whatever envelope the caller passes, the
        matrices use Envelope(seqPair) -- the path envelope of an aligned pair, else the full one."""
        b = self.deviceBatch(self.machine, seqPairs)
        _, s, ll = b.counts(self._flat)
        b.close()
        self.loglike += s
        return [float(x) for x in ll]

    def __iadd__(self, other: "MachineCounts") -> "MachineCounts":
        self._flat += other._flat
        self.loglike += other.loglike
        return self

    def paramCounts(self, m: Machine, prob: Dict[str, Any]) -> Dict[str, float]:
        """src/counts.cpp:88-106: sum over transitions of  count * (dw/dp) * p / w.  Like the reference, the parameters of
        a weight are the names that appear IN it (WeightAlgebra::params / deriv are called with empty ParamDefs: function
        definitions are not expanded), and values come from ``prob`` on top of the machine's own defs."""
        defs = dict(m.funcs); defs.update(prob)
        out: Dict[str, float] = {}
        e = 0
        for s, ms in enumerate(m.state):
            for t in ms.trans:
                c = float(self._flat[e]); e += 1
                params = _freeParams(t.weight, {})
                if not params:
                    continue
                w = evalWeight(t.weight, defs)
                for p in sorted(params):
                    d = _deriv(t.weight, defs, p, expand=False)
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


def _deriv(w: Any, defs: Dict[str, Any], p: str, expand: bool = True) -> float:
    """d eval(w) / d p, forward-mode over the JSON expression (the reference differentiates symbolically,
    src/weight.cpp:302-380, then evaluates; the value is the same).  expand=False treats every other name as a constant
    (WeightAlgebra::deriv with empty ParamDefs); values always come from ``defs``."""
    if w is None or isinstance(w, (bool, int, float)):
        return 0.0
    if isinstance(w, str):
        if w == p:
            return 1.0
        v = defs.get(w)
        if not expand or v is None or isinstance(v, (int, float)):
            return 0.0
        return _deriv(v, {k: x for k, x in defs.items() if k != w}, p)
    op, args = next(iter(w.items()))
    ev = lambda x: evalWeight(x, defs)
    if op == "log":
        return _deriv(args, defs, p, expand) / ev(args)
    if op == "exp":
        return _deriv(args, defs, p, expand) * math.exp(ev(args))
    if op == "not":
        return -_deriv(args, defs, p, expand)
    if op == "geomsum":
        return _deriv(args, defs, p, expand) / (1.0 - ev(args)) ** 2
    a, b = args
    if op == "*":
        return _deriv(a, defs, p, expand) * ev(b) + ev(a) * _deriv(b, defs, p, expand)
    if op == "/":
        return (_deriv(a, defs, p, expand) * ev(b) - ev(a) * _deriv(b, defs, p, expand)) / ev(b) ** 2
    if op == "+":
        return _deriv(a, defs, p, expand) + _deriv(b, defs, p, expand)
    if op == "-":
        return _deriv(a, defs, p, expand) - _deriv(b, defs, p, expand)
    raise MachineError("Unknown opcode %s" % op)

"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY -- see mb_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SUM_TABLE, SUM_EXACT, MAX = 0, 1, 2


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libmboracle.so")
    src = os.path.join(_HERE, "mb_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32p, u32p, u16p, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_double)
        L.mbo_init.restype = None
        L.mbo_log_sum_exp.restype = C.c_double
        L.mbo_log_sum_exp.argtypes = [C.c_double, C.c_double, C.c_int]
        L.mbo_machine_create.restype = vp
        L.mbo_machine_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, u32p, u32p, u16p, u16p, dp]
        L.mbo_machine_set_weights.argtypes = [vp, dp]
        L.mbo_machine_destroy.argtypes = [vp]
        L.mbo_incoming_order.argtypes = [vp, u32p]
        L.mbo_outgoing_order.argtypes = [vp, u32p]
        L.mbo_fill_forward.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, C.c_int, C.c_int, dp]
        L.mbo_fill_backward.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, C.c_int, dp]
        L.mbo_forward_loglike.restype = C.c_double
        L.mbo_forward_loglike.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, C.c_int]
        L.mbo_get_counts.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, dp, dp, dp]
        L.mbo_counts_add.restype = C.c_double
        L.mbo_counts_add.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, C.c_int, dp]
        L.mbo_traceback.restype = C.c_long
        L.mbo_traceback.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, dp, u32p, C.c_long]
        L.mbo_set_envelope.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.mbo_set_envelope.restype = None
        u8p, lp = C.POINTER(C.c_uint8), C.POINTER(C.c_long)
        L.mbo_mt_create.restype = vp; L.mbo_mt_create.argtypes = [C.c_uint32]
        L.mbo_mt_destroy.argtypes = [vp]
        L.mbo_mt_next.restype = C.c_uint32; L.mbo_mt_next.argtypes = [vp]
        L.mbo_set_result_bits.argtypes = [C.c_int]; L.mbo_set_result_bits.restype = None
        for fn in (L.mbo_trace_back, L.mbo_trace_forward):
            fn.restype = C.c_long
            fn.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, dp, C.c_long, C.c_long, C.c_int, C.c_int, vp, u8p, u32p, C.c_long]
        L.mbo_post_trans.restype = C.c_long
        L.mbo_post_trans.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, dp, dp, lp, lp, u32p, dp, C.c_long]
        L.mbo_trace_from.restype = C.c_long
        L.mbo_trace_from.argtypes = [vp, i32p, C.c_long, i32p, C.c_long, dp, dp, C.c_long, C.c_long, C.c_uint32, u8p, u32p, C.c_long]
        L.mbo_init()
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class envelope:
    """Context manager: fills inside the block use the given envelope (inStart[o], inEnd[o]); None = full."""

    def __init__(self, inStart=None, inEnd=None):
        self.a = None if inStart is None else np.ascontiguousarray(inStart, np.int64)
        self.b = None if inEnd is None else np.ascontiguousarray(inEnd, np.int64)

    def __enter__(self):
        if self.a is not None:
            lib().mbo_set_envelope(_p(self.a, C.c_long), _p(self.b, C.c_long))
        return self

    def __exit__(self, *exc):
        lib().mbo_set_envelope(None, None)


class OracleMachine:
    """Oracle-side flattened machine built from a machineboss_amd.evalmachine.EvaluatedMachine."""

    def __init__(self, em):
        self.em = em
        self.L = lib()
        self._keep = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
                      np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
                      np.ascontiguousarray(em.logWeight, np.float64)]
        k = self._keep
        self.h = self.L.mbo_machine_create(em.nStates, em.nInTok, em.nOutTok, em.nTransitions,
                                           _p(k[0], C.c_uint32), _p(k[1], C.c_uint32), _p(k[2], C.c_uint16),
                                           _p(k[3], C.c_uint16), _p(k[4], C.c_double))
        self.S = em.nStates
        self.nT = em.nTransitions

    def __del__(self):
        try:
            self.L.mbo_machine_destroy(self.h)
        except Exception:
            pass

    def set_weights(self, lw):
        lw = np.ascontiguousarray(lw, np.float64)
        self.L.mbo_machine_set_weights(self.h, _p(lw, C.c_double))

    def incoming_order(self):
        o = np.empty(self.nT, np.uint32); self.L.mbo_incoming_order(self.h, _p(o, C.c_uint32)); return o

    def outgoing_order(self):
        o = np.empty(self.nT, np.uint32); self.L.mbo_outgoing_order(self.h, _p(o, C.c_uint32)); return o

    @staticmethod
    def _seqs(inp, out):
        return np.ascontiguousarray(inp, np.int32), np.ascontiguousarray(out, np.int32)

    def forward(self, inp, out, mode=SUM_TABLE, startState=0):
        """Full Forward (mode SUM_*) or Viterbi (mode MAX) matrix, shape [outLen+1][inLen+1][nStates]."""
        i, o = self._seqs(inp, out)
        cells = np.empty((len(o) + 1, len(i) + 1, self.S), np.float64)
        self.L.mbo_fill_forward(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), mode, startState, _p(cells, C.c_double))
        return cells

    def viterbi(self, inp, out):
        return self.forward(inp, out, MAX, 0)

    def backward(self, inp, out, mode=SUM_TABLE):
        i, o = self._seqs(inp, out)
        cells = np.empty((len(o) + 1, len(i) + 1, self.S), np.float64)
        self.L.mbo_fill_backward(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), mode, _p(cells, C.c_double))
        return cells

    def loglike(self, inp, out, mode=SUM_TABLE):
        i, o = self._seqs(inp, out)
        return self.L.mbo_forward_loglike(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), mode)

    def counts_add(self, inp, out, counts, mode=SUM_TABLE):
        i, o = self._seqs(inp, out)
        assert counts.dtype == np.float64 and counts.shape == (self.nT,)
        return self.L.mbo_counts_add(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), mode, _p(counts, C.c_double))

    def traceback(self, inp, out, cells):
        i, o = self._seqs(inp, out)
        cap = (len(i) + len(o) + 2) * (self.S + 1)
        path = np.empty(cap, np.uint32)
        n = self.L.mbo_traceback(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), _p(cells, C.c_double), _p(path, C.c_uint32), cap)
        if n < 0:
            raise RuntimeError("Can't do traceback: no finite-weight paths" if n == -1 else "traceback error %d" % n)
        return path[:n].copy()

    # ---- the matrix walkers (src/dpmatrix.defs.h:61-186, src/backward.cpp:52-108, src/forward.cpp:17-23) -------------
    def _walk(self, fn, inp, out, cells, inPos, outPos, s, rng, mask):
        i, o = self._seqs(inp, out)
        cells = np.ascontiguousarray(cells, np.float64)
        cap = (len(i) + len(o) + 2) * (self.nT + 2)
        edges = np.empty(cap, np.uint32)
        n = fn(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), _p(cells, C.c_double), inPos, outPos, s,
               0 if rng is None else 1, None if rng is None else rng.h, None if mask is None else _p(mask, C.c_uint8), _p(edges, C.c_uint32), cap)
        if n < 0:
            raise RuntimeError("Can't do traceback: no finite-weight paths" if n == -1 else "walk error %d" % n)
        return edges[:n].copy()

    def trace_back(self, inp, out, cells, inPos=None, outPos=None, s=None, rng=None, mask=None):
        """DPMatrix::traceBack with a terminator (mask: Machine::downsample's) and selectMaxTrans or, with ``rng`` (an
        Mt19937), randomTransSelector.  Edge ids in the order the steps are taken (end -> start)."""
        return self._walk(self.L.mbo_trace_back, inp, out, cells, len(inp) if inPos is None else inPos,
                          len(out) if outPos is None else outPos, self.S - 1 if s is None else s, rng, mask)

    def trace_forward(self, inp, out, cells, inPos=0, outPos=0, s=0, rng=None, mask=None):
        return self._walk(self.L.mbo_trace_forward, inp, out, cells, inPos, outPos, s, rng, mask)

    def post_trans(self, inp, out, fwd, bwd):
        """BackwardMatrix::getCounts through a visitor: (inPos, outPos, edge, weight) arrays in visit order."""
        i, o = self._seqs(inp, out)
        cap = (len(i) + 1) * (len(o) + 1) * max(self.nT, 1) + 1
        ip = np.empty(cap, np.int64); op = np.empty(cap, np.int64); e = np.empty(cap, np.uint32); w = np.empty(cap, np.float64)
        fwd = np.ascontiguousarray(fwd, np.float64); bwd = np.ascontiguousarray(bwd, np.float64)
        n = self.L.mbo_post_trans(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), _p(fwd, C.c_double), _p(bwd, C.c_double),
                                  _p(ip, C.c_long), _p(op, C.c_long), _p(e, C.c_uint32), _p(w, C.c_double), cap)
        assert n >= 0
        return ip[:n].copy(), op[:n].copy(), e[:n].copy(), w[:n].copy()

    def trace_from(self, inp, out, fwd, bwd, inPos, outPos, edge, mask=None):
        """BackwardMatrix::traceFrom with a terminator: the transition, the traceback from its source, the traceforward
        from its destination (edge ids in visit order)."""
        i, o = self._seqs(inp, out)
        fwd = np.ascontiguousarray(fwd, np.float64); bwd = np.ascontiguousarray(bwd, np.float64)
        cap = 2 * (len(i) + len(o) + 2) * (self.nT + 2) + 1
        edges = np.empty(cap, np.uint32)
        n = self.L.mbo_trace_from(self.h, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), _p(fwd, C.c_double), _p(bwd, C.c_double),
                                  inPos, outPos, int(edge), None if mask is None else _p(mask, C.c_uint8), _p(edges, C.c_uint32), cap)
        if n < 0:
            raise RuntimeError("traceFrom error %d" % n)
        return edges[:n].copy()


def set_result_bits(bits: int):
    """Width of std::mt19937::result_type assumed by random_double: 64 (libstdc++ on LP64 Linux, the default -- see
    mb_oracle.c) or 32 (libc++)."""
    lib().mbo_set_result_bits(bits)


class Mt19937:
    """std::mt19937(seed) of the oracle (mbo_mt_*)."""

    def __init__(self, seed=5489):
        self.L = lib(); self.h = self.L.mbo_mt_create(int(seed) & 0xFFFFFFFF)

    def __call__(self):
        return int(self.L.mbo_mt_next(self.h))

    @staticmethod
    def max():
        return 0xFFFFFFFF

    def __del__(self):
        try:
            self.L.mbo_mt_destroy(self.h)
        except Exception:
            pass

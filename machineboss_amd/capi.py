"""ctypes binding of the C-ABI in include/mbhip.h (libmbhip.so, built in-tree by machineboss_amd.build).

This is the stub a Python caller of the reference (python/machineboss/boss.py shells out to the ``boss`` CLI) would
use instead.  There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MBHIP_LIBRARY") or os.path.join(HERE, "libmbhip.so")      # (MBHIP_LIBRARY: another build of the same library, e.g. the sanitizer build of scripts/asan_planner.sh)

MB_FORWARD, MB_VITERBI, MB_BACKWARD = 0, 1, 2
MB_MATERIALISE, MB_ROLLING = 0, 1
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SMALL, KERNEL_MEDIUM = 0, 1, 2, 3

EXPORTS = [
    "mb_device_count", "mb_set_device", "mb_synchronize", "mb_last_error", "mb_last_device_ms", "mb_last_kernel_name", "mb_last_launch_count",
    "mb_machine_create", "mb_machine_set_weights", "mb_machine_destroy", "mb_machine_n_states", "mb_machine_n_trans",
    "mb_machine_n_levels", "mb_machine_edge_order",
    "mb_batch_create", "mb_batch_destroy", "mb_batch_cells", "mb_batch_forward", "mb_viterbi_path_bound",
    "mb_batch_viterbi", "mb_batch_counts", "mb_fill", "mb_forward_batch", "mb_viterbi_batch", "mb_counts_batch",
    "mb_set_kernel", "mb_set_memory_budget", "mb_release_workspace", "mb_debug_jit_source", "mb_debug_small_source", "mb_debug_wide_retimed", "mb_debug_wide_parts", "mb_debug_wide_jit",
    "mb_jit_stats", "mb_alloc_stats", "mb_machine_sweep_ops", "mb_set_option", "mb_get_option", "mb_log_sum_exp", "mb_log_sum_exp_n", "mb_log_inner_product",
    "mb_batch_set_envelopes", "mb_fill_env",
    "mb_comm_unique_id", "mb_comm_init", "mb_comm_destroy", "mb_allreduce_counts",
]

_lib = None


class MbError(RuntimeError):
    """The library's non-zero status, carrying mb_last_error() -- the reference throws runtime_error (src/util.cpp:39-48)."""


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MbError("libmbhip.so is not built (run `python -m machineboss_amd.build`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    i32p, i64p, u32p, u16p, dp = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint16), C.POINTER(C.c_double))
    L.mb_device_count.restype = C.c_int
    L.mb_set_device.argtypes = [C.c_int]
    L.mb_last_error.restype = C.c_char_p
    L.mb_last_device_ms.restype = C.c_double
    L.mb_last_kernel_name.restype = C.c_char_p
    L.mb_last_launch_count.restype = C.c_int64
    L.mb_machine_create.restype = vp
    L.mb_machine_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp]
    L.mb_machine_set_weights.argtypes = [vp, dp]
    L.mb_machine_destroy.argtypes = [vp]
    L.mb_machine_destroy.restype = None
    L.mb_machine_n_states.argtypes = [vp]; L.mb_machine_n_states.restype = C.c_int32
    L.mb_machine_n_trans.argtypes = [vp]; L.mb_machine_n_trans.restype = C.c_int64
    L.mb_machine_n_levels.argtypes = [vp]; L.mb_machine_n_levels.restype = C.c_int32
    L.mb_machine_edge_order.argtypes = [vp, C.c_int, u32p]
    L.mb_batch_create.restype = vp
    L.mb_batch_create.argtypes = [vp, C.c_int64, i32p, i64p, i32p, i64p]
    L.mb_batch_destroy.argtypes = [vp]; L.mb_batch_destroy.restype = None
    L.mb_batch_cells.argtypes = [vp]; L.mb_batch_cells.restype = C.c_int64
    L.mb_batch_forward.argtypes = [vp, C.c_int, dp]
    L.mb_viterbi_path_bound.argtypes = [vp, C.c_int64, C.c_int64]; L.mb_viterbi_path_bound.restype = C.c_int64
    L.mb_batch_viterbi.argtypes = [vp, dp, i64p, u32p, C.c_int64]
    L.mb_batch_counts.argtypes = [vp, dp, dp, dp]
    L.mb_fill.argtypes = [vp, C.c_int, i32p, C.c_int64, i32p, C.c_int64, C.c_int32, dp]
    L.mb_forward_batch.argtypes = [vp, C.c_int64, i32p, i64p, i32p, i64p, C.c_int, dp]
    L.mb_viterbi_batch.argtypes = [vp, C.c_int64, i32p, i64p, i32p, i64p, dp, i64p, u32p, C.c_int64]
    L.mb_counts_batch.argtypes = [vp, C.c_int64, i32p, i64p, i32p, i64p, dp, dp, dp]
    L.mb_set_kernel.argtypes = [C.c_int]
    L.mb_set_memory_budget.argtypes = [C.c_size_t]
    L.mb_batch_set_envelopes.argtypes = [vp, i64p, i32p, i32p]
    L.mb_fill_env.argtypes = [vp, C.c_int, i32p, C.c_int64, i32p, C.c_int64, C.c_int32, i32p, i32p, dp]
    L.mb_debug_jit_source.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.mb_debug_small_source.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp,
                                        C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.mb_debug_wide_retimed.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp,
                                        C.c_int, C.c_int, C.c_char_p]
    L.mb_debug_wide_parts.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.mb_debug_wide_jit.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, u32p, u32p, u16p, u16p, dp,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.mb_log_sum_exp.argtypes = [C.c_double, C.c_double]; L.mb_log_sum_exp.restype = C.c_double
    L.mb_log_sum_exp_n.argtypes = [dp, C.c_size_t]; L.mb_log_sum_exp_n.restype = C.c_double
    L.mb_log_inner_product.argtypes = [dp, dp, dp, C.c_size_t]; L.mb_log_inner_product.restype = C.c_double
    L.mb_jit_stats.argtypes = [dp, i64p, i64p]
    L.mb_alloc_stats.argtypes = [i64p, i64p, i64p, C.POINTER(C.c_uint64), dp]
    L.mb_machine_sweep_ops.argtypes = [C.c_void_p, dp, dp, C.POINTER(C.c_char_p)]
    L.mb_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.mb_get_option.argtypes = [C.c_char_p]; L.mb_get_option.restype = C.c_char_p
    L.mb_comm_unique_id.argtypes = [C.c_char_p]
    L.mb_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int]; L.mb_comm_init.restype = vp
    L.mb_comm_destroy.argtypes = [vp]; L.mb_comm_destroy.restype = None
    L.mb_allreduce_counts.argtypes = [vp, dp, C.c_size_t, dp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MbError(load().mb_last_error().decode() or "mbhip error")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def device_count() -> int:
    return load().mb_device_count()


def set_device(d: int):
    _check(load().mb_set_device(d))


def set_kernel(which: int):
    _check(load().mb_set_kernel(which))


def set_memory_budget(nbytes: int):
    _check(load().mb_set_memory_budget(nbytes))


def release_workspace():
    _check(load().mb_release_workspace())


def jit_stats() -> dict:
    """hiprtc work done by this process: compile wall time, number of compiles, code objects served by the disk cache."""
    ms = C.c_double(0.0); n = C.c_int64(0); h = C.c_int64(0)
    _check(load().mb_jit_stats(C.byref(ms), C.byref(n), C.byref(h)))
    return {"compile_ms": round(ms.value, 1), "compiles": n.value, "cache_hits": h.value}


def alloc_stats() -> dict:
    """Device allocations the matrix pools cost this process so far (steady-state calls do none)."""
    a = C.c_int64(0); f = C.c_int64(0); e = C.c_int64(0); b = C.c_uint64(0); ms = C.c_double(0.0)
    _check(load().mb_alloc_stats(C.byref(a), C.byref(f), C.byref(e), C.byref(b), C.byref(ms)))
    return {"pool_allocs": a.value, "pool_frees": f.value, "evictions": e.value, "bytes_allocated": b.value, "ms": round(ms.value, 2)}


def sweep_ops(dm) -> dict:
    """v_exp_f32 / v_log_f32 per lattice cell of the machine's log-sum-exp Forward sweep, and the kernel family that runs it."""
    e = C.c_double(0.0); l = C.c_double(0.0); f = C.c_char_p()
    _check(load().mb_machine_sweep_ops(dm.h, C.byref(e), C.byref(l), C.byref(f)))
    return {"exp_per_cell": e.value, "log_per_cell": l.value, "family": (f.value or b"").decode()}


def set_option(name: str, value=None):
    _check(load().mb_set_option(name.encode(), None if value is None else str(value).encode()))


def last_device_ms() -> float:
    return load().mb_last_device_ms()


def last_launch_count() -> int:
    return load().mb_last_launch_count()


def last_kernel_name() -> str:
    return load().mb_last_kernel_name().decode()


def synchronize():
    """Wait for everything the library queued on its device (the timing bracket of a multi-rank host)."""
    _check(load().mb_synchronize())


class Comm:
    """RCCL communicator of the C-ABI (mb_comm*) on the library's own HIP runtime and stream: what a C++ host uses, and what
    shard.RankGroup opens for the Python hosts (bench.py, boss.py)."""

    def __init__(self, unique_id: bytes, nRanks: int, rank: int):
        self.h = load().mb_comm_init(unique_id, nRanks, rank)
        if not self.h:
            raise MbError(load().mb_last_error().decode())

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().mb_comm_unique_id(buf))
        return buf.raw

    def allreduce_counts(self, counts: np.ndarray, loglike: float):
        assert counts.dtype == np.float64 and counts.flags.c_contiguous
        ll = C.c_double(loglike)
        _check(load().mb_allreduce_counts(self.h, _p(counts, C.c_double), counts.size, C.byref(ll)))
        return counts, ll.value

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mb_comm_destroy(self.h)
            self.h = None


def debug_jit_source(em, path: str, mode: int = MB_FORWARD, backward: bool = False, closure: int = 1, G: int = 2):
    """Write the HIP source of the run-time specialised tile kernel for this machine (host only, no GPU needed).
    closure: 0 = levelled program, K >= 1 = silent closure in K stages."""
    a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
         np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
         np.ascontiguousarray(em.logWeight, np.float64)]
    _check(load().mb_debug_jit_source(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                      _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double),
                                      mode, int(backward), int(closure), G, path.encode()))


def debug_medium_program(em, path: str, mode: int = MB_FORWARD, backward: bool = False, closure: int = 1, G: int = 2) -> dict:
    """The tiled family's PROGRAM for this machine as the kernels read it (host only, no GPU needed): chunk descriptors `desc`
    [nChunks][8], records `rec` (w, srcOff, dstOff), the usage slots `flat` [(table, first record)] of a flat count program, and
    the scalars S, Spad, LPG, G, nIn, nOut, seedOff, dummyOff, backward, closure, counting, flatCount, fusedEmit, twoTables, the fused emit slots
    `fused` and the loop-time accumulator map `accMap` of a flat count program (mb_api.hip,
    mb_debug_jit_source with mode + 32; the semantics are med_slow_supercell's, mb_medium.hip)."""
    debug_jit_source(em, path, mode=mode | 32, backward=backward, closure=closure, G=G)
    head = np.fromfile(path, np.int32, 16)
    assert head[0] == 0x4D454431
    keys = ("S", "Spad", "LPG", "G", "nChunks", "nIn", "nOut", "seedOff", "dummyOff", "nRec", "nFlat", "backward", "closure", "counting", "flatCount")
    out = {k: int(v) for k, v in zip(keys, head[1:])}
    off = 64
    out["desc"] = np.fromfile(path, np.int32, out["nChunks"] * 8, offset=off).reshape(out["nChunks"], 8); off += out["nChunks"] * 32
    out["rec"] = np.fromfile(path, np.dtype([("w", "<f8"), ("srcOff", "<u4"), ("dstOff", "<u4")]), out["nRec"], offset=off); off += out["nRec"] * 16
    flags = out["flatCount"]
    out["flatCount"], out["fusedEmit"], out["twoTables"] = flags & 1, (flags >> 1) & 1, (flags >> 2) & 1
    out["flat"] = np.fromfile(path, np.int32, out["nFlat"] * 3, offset=off).reshape(out["nFlat"], 3); off += out["nFlat"] * 12      # (table, first record, placement: 2 = VGPRs)
    out["wref"] = np.fromfile(path, np.int32, out["nRec"], offset=off); off += out["nRec"] * 4      # per record: >= 0 its transition, -1 padding, <= -2 a closure pair
    nFused, nLoop = (int(v) for v in np.fromfile(path, np.int32, 2, offset=off)); off += 8
    out["fused"] = np.fromfile(path, np.int32, nFused * 3, offset=off).reshape(nFused, 3); off += nFused * 12                        # emit slots of the fill rounds that also add usage
    out["accMap"] = np.fromfile(path, np.int32, nLoop, offset=off); off += nLoop * 4                                                  # loop-time accumulator entry -> transition
    assert os.path.getsize(path) == off
    return out


def debug_small_source(em, path: str, mode: int = 0, backward: bool = False, materialise: bool = True, envelopes: bool = False):
    """Write the HIP source of the small-machine family's sweep for this machine (host only, no GPU needed).
    mode: 0 sum, 1 max (fp64 cells), 2 max (traceback bytes), 3 Forward fused with posterior counts."""
    a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
         np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
         np.ascontiguousarray(em.logWeight, np.float64)]
    _check(load().mb_debug_small_source(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                        _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double),
                                        mode, int(backward), int(materialise) | (2 if envelopes else 0), path.encode()))


def debug_small_program(em, path: str, backward: bool = False) -> dict:
    """The small-machine family's PROGRAM for this machine (host only, no GPU needed; mb_debug_small_source with mode + 32): the
    evaluation order of the states, per state its candidates (T, src, dup, tab) in the reference's enumeration order (`decOff` [S + 1]
    delimits them), and the weight / edge-id tables `w`, `eid` with their layout (`off`, `nTab` by kind T: 0 match, 1 input-only,
    2 output-only, 3 silent)."""
    debug_small_source(em, path, mode=32, backward=backward)
    head = np.fromfile(path, np.int32, 20)
    assert head[0] == 0x534D5031
    out = {k: int(v) for k, v in zip(("S", "nIn", "nOut", "backward", "seedState", "endState", "nEntries"), head[1:8])}
    out["off"] = [int(v) for v in head[8:12]]; out["nTab"] = [int(v) for v in head[12:16]]
    nc, S, pos = int(head[16]), out["S"], 80
    out["order"] = np.fromfile(path, np.int32, S, offset=pos); pos += 4 * S
    out["decOff"] = np.fromfile(path, np.int32, S + 1, offset=pos); pos += 4 * (S + 1)
    out["cand"] = np.fromfile(path, np.int32, 4 * nc, offset=pos).reshape(nc, 4); pos += 16 * nc
    out["w"] = np.fromfile(path, np.float64, out["nEntries"], offset=pos); pos += 8 * out["nEntries"]
    out["eid"] = np.fromfile(path, np.int32, out["nEntries"], offset=pos); pos += 4 * out["nEntries"]
    assert os.path.getsize(path) == pos and out["decOff"][-1] == nc
    return out


def debug_wide_retimed(em, path: str, mode: int = MB_VITERBI, backward: bool = False, tb_codes: bool = False) -> dict:
    """The retimed program of a one-tape machine as the kernel reads it (host only, no GPU needed): the header fields and
    the record streams, one per rotation of the ring, as a structured array [stream][slot][lane] of (w, src, pad).
    tb_codes (forward max program only): the program that keeps one traceback code per cell, with its decode tables
    `tbOff` [S + 1], `tbEntry` (position in the incoming view << 16 | emitting << 15 | source state; 0xFFFFFFFF: the seed)
    and `inEid` (incoming view position -> global edge id)."""
    a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
         np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
         np.ascontiguousarray(em.logWeight, np.float64)]
    _check(load().mb_debug_wide_retimed(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                        _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double),
                                        mode | (16 if tb_codes else 0), int(backward), path.encode()))
    head = np.fromfile(path, np.int32, 12)
    assert head[0] == 0x52455431
    keys = ("lanes", "slots", "NB", "NVs", "kMax", "rowLen", "nPen", "period", "S", "inL2", "streams")
    out = {k: int(v) for k, v in zip(keys, head[1:])}
    nrec = (out["NB"] * out["slots"] + 8) * out["lanes"]
    rec = np.fromfile(path, np.dtype([("w", "<f8"), ("src", "<u4"), ("pad", "<u4")]), count=nrec, offset=48)
    assert rec.size == nrec
    out["records"] = rec[:out["NB"] * out["slots"] * out["lanes"]].reshape(out["NB"], out["slots"], out["lanes"])
    out["tail"] = rec[out["NB"] * out["slots"] * out["lanes"]:].reshape(8, out["lanes"])
    rest = np.fromfile(path, np.uint32, offset=48 + 16 * nrec)
    if tb_codes:
        pos = 0
        for k in ("tbOff", "tbEntry", "inEid"):
            n = int(rest[pos]); out[k] = rest[pos + 1:pos + 1 + n].copy(); pos += 1 + n
        assert pos == rest.size and out["tbOff"].size == em.nStates + 1 and out["inEid"].size == em.nTransitions
    else:
        assert rest.size == 0
    return out


def debug_wide_parts(em, path: str, k: int, lanes: int = 256, mode: int = MB_VITERBI, backward: bool = False, tb_codes: bool = False) -> dict:
    """The k-part form of the retimed program (k workgroups per sequence; host only): {"nExp": exchange columns, "S": states,
    "parts": [per part the fields of debug_wide_retimed + Sloc, nImp, expBase, expIdx0, nExp, resultEntry, gmap, impIdx]}; tb_codes:
    + the decode tables of the parts' candidate lists, joined over the machine's states (tbOff, tbEntry, inEid)."""
    a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
         np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
         np.ascontiguousarray(em.logWeight, np.float64)]
    _check(load().mb_debug_wide_parts(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                      _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double),
                                      mode | (16 if tb_codes else 0), int(backward), int(k), int(lanes), path.encode()))
    head = np.fromfile(path, np.int32, 4)
    assert head[0] == 0x52455432
    out = {"nExp": int(head[2]), "S": int(head[3]), "parts": []}
    pos = 16
    keys = ("lanes", "slots", "NB", "NVs", "kMax", "rowLen", "nPen", "period", "Sloc", "nImp", "expBase", "expIdx0", "nExp", "resultEntry", "nTab", "w2Offset")
    for _ in range(int(head[1])):
        ph = np.fromfile(path, np.int32, 16, offset=pos); pos += 64
        part = {k_: int(v) for k_, v in zip(keys, ph)}
        tab = np.fromfile(path, np.uint32, part["nTab"], offset=pos); pos += 4 * part["nTab"]
        part["gmap"], part["impIdx"] = tab[:part["Sloc"]].astype(np.int64), tab[part["Sloc"]:].astype(np.int64)
        assert part["impIdx"].size == part["nImp"]
        nrec = (part["NB"] * part["slots"] + 8) * part["lanes"]
        rec16 = np.fromfile(path, np.dtype([("w", "<f8"), ("src", "<u4"), ("pad", "<u4")]), count=nrec, offset=pos); pos += 16 * nrec
        rec = np.zeros(nrec, np.dtype([("w", "<f8"), ("src", "<u4"), ("pad", "<u4"), ("w2", "<f8")]))
        for f in ("w", "src", "pad"): rec[f] = rec16[f]
        if part["w2Offset"]:      # two-transition candidates: the second weights, one per record, behind the records
            assert part["w2Offset"] == 16 * nrec
            rec["w2"] = np.fromfile(path, "<f8", count=nrec, offset=pos); pos += 8 * nrec
        part["records"] = rec[:part["NB"] * part["slots"] * part["lanes"]].reshape(part["NB"], part["slots"], part["lanes"])
        part["inL2"] = 0
        out["parts"].append(part)
    if tb_codes:
        rest = np.fromfile(path, np.uint32, offset=pos); q = 0
        for k_ in ("tbOff", "tbEntry", "inEid"):
            n = int(rest[q]); out[k_] = rest[q + 1:q + 1 + n].copy(); q += 1 + n
        assert q == rest.size and out["tbOff"].size == em.nStates + 1 and out["inEid"].size == em.nTransitions
        pos += 4 * rest.size
    assert os.path.getsize(path) == pos
    return out


def debug_wide_jit(em, path: str, k: int = 1, lanes: int = 0, mode: int = MB_VITERBI, backward: bool = False, tb_codes: bool = False,
                   acc: bool = False, compile: bool = False) -> dict:
    """The one-tape sweep GENERATED for this machine (mb_wide_jit.cpp; host only): the HIP source goes to `path`, and what it unrolls
    comes back as {"nExp", "S", "parts": [{geometry ..., "slots": [(anyPen, anyW2)], "rounds": [{firstSlot, depth, sync, uniform, gAll,
    anyMixed, resultLane}], "fields": [(kind, index, cm, words)], "table": uint32 [words][lanes], "impIdx"}]} -- the per-lane constant
    table exactly as the kernel loads it.  compile: also through hiprtc (raises when the kernel would spill to scratch memory)."""
    a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
         np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
         np.ascontiguousarray(em.logWeight, np.float64)]
    _check(load().mb_debug_wide_jit(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                    _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double),
                                    mode | (16 if tb_codes else 0) | (64 if acc else 0), int(backward), int(k), int(lanes), int(compile), path.encode()))
    prog = path + ".prog"
    head = np.fromfile(prog, np.int32, 4)
    assert head[0] == 0x4A495431
    out = {"nExp": int(head[2]), "S": int(head[3]), "parts": []}
    pos = 16
    keys = ("lanes", "NB", "NVs", "kMax", "rowLen", "nPen", "nImp", "Sloc", "expBase", "nExp", "expIdx0", "resultEntry", "nSlots", "nRounds", "NPT", "U",
            "penBase", "tokBase", "ringBase", "dummyAddr", "ldsBytes", "nFields", "nWords", "pad")
    for _ in range(int(head[1])):
        ph = np.fromfile(prog, np.int32, 24, offset=pos); pos += 96
        part = {k_: int(v) for k_, v in zip(keys, ph)}
        sl = np.fromfile(prog, np.int32, 2 * part["nSlots"], offset=pos).reshape(-1, 2); pos += 8 * part["nSlots"]
        rd = np.fromfile(prog, np.int32, 8 * part["nRounds"], offset=pos).reshape(-1, 8); pos += 32 * part["nRounds"]
        fd = np.fromfile(prog, np.int32, 4 * part["nFields"], offset=pos).reshape(-1, 4); pos += 16 * part["nFields"]
        part["slots"] = [(bool(x[0]), bool(x[1])) for x in sl]
        part["rounds"] = [dict(firstSlot=int(x[0]), depth=int(x[1]), sync=bool(x[2]), uniform=bool(x[3]), gAll=int(x[4]), anyMixed=bool(x[5]), resultLane=int(x[6]), gMask=int(x[7])) for x in rd]
        part["fields"] = [tuple(int(v) for v in x) for x in fd]
        part["table"] = np.fromfile(prog, np.uint32, part["nWords"] * part["lanes"], offset=pos).reshape(part["nWords"], part["lanes"]); pos += 4 * part["nWords"] * part["lanes"]
        part["level"], part["IP"], part["ring"] = part["pad"] & 0xff, (part["pad"] >> 8) & 0xfff, part["pad"] >> 20
        if part["level"] >= 1:      # the packed address words of a streamed program: [rotation][item][lane]
            n = part["NB"] * part["IP"] * part["lanes"]
            part["stream"] = np.fromfile(prog, np.uint32, n, offset=pos).reshape(part["NB"], part["IP"], part["lanes"]); pos += 4 * n
        if len(out["parts"]) > 0 or int(head[1]) > 1:
            part["impIdx"] = np.fromfile(prog, np.uint32, part["nImp"], offset=pos).astype(np.int64); pos += 4 * part["nImp"]
        out["parts"].append(part)
    assert os.path.getsize(prog) == pos
    return out


class DeviceMachine:
    """Device-resident flattened machine (mb_machine*), built from an evalmachine.EvaluatedMachine."""

    def __init__(self, em):
        L = load()
        self.em = em
        a = [np.ascontiguousarray(em.src, np.uint32), np.ascontiguousarray(em.dst, np.uint32),
             np.ascontiguousarray(em.inTok, np.uint16), np.ascontiguousarray(em.outTok, np.uint16),
             np.ascontiguousarray(em.logWeight, np.float64)]
        self.h = L.mb_machine_create(em.nStates, em.nInTok, em.nOutTok, em.nTransitions, _p(a[0], C.c_uint32),
                                     _p(a[1], C.c_uint32), _p(a[2], C.c_uint16), _p(a[3], C.c_uint16), _p(a[4], C.c_double))
        if not self.h:
            raise MbError(L.mb_last_error().decode())
        self.nStates, self.nTrans = em.nStates, em.nTransitions

    def close(self):
        if getattr(self, "h", None) and _lib is not None:   # at interpreter shutdown the module globals may be gone
            try:
                _lib.mb_machine_destroy(self.h)
            except Exception:
                pass
            self.h = None

    __del__ = close

    def set_weights(self, logWeight):
        lw = np.ascontiguousarray(logWeight, np.float64)
        assert lw.shape == (self.nTrans,)
        _check(load().mb_machine_set_weights(self.h, _p(lw, C.c_double)))

    def n_levels(self) -> int:
        return load().mb_machine_n_levels(self.h)

    def edge_order(self, which: int) -> np.ndarray:
        o = np.empty(self.nTrans, np.uint32)
        _check(load().mb_machine_edge_order(self.h, which, _p(o, C.c_uint32)))
        return o

    def fill(self, mode: int, inp, out, startState: int = 0, envStart=None, envEnd=None) -> np.ndarray:
        """Full matrix [outLen+1][inLen+1][nStates] (DPMatrix cell storage, src/dpmatrix.h:90-96); with an envelope
        (inStart[o], inEnd[o] per output position) the cells outside it are -inf."""
        i = np.ascontiguousarray(inp, np.int32); o = np.ascontiguousarray(out, np.int32)
        cells = np.empty((len(o) + 1, len(i) + 1, self.nStates), np.float64)
        es = None if envStart is None else np.ascontiguousarray(envStart, np.int32)
        ee = None if envEnd is None else np.ascontiguousarray(envEnd, np.int32)
        _check(load().mb_fill_env(self.h, mode, _p(i, C.c_int32), len(i), _p(o, C.c_int32), len(o), startState,
                                  _p(es, C.c_int32), _p(ee, C.c_int32), _p(cells, C.c_double)))
        return cells

    def path_bound(self, inLen: int, outLen: int) -> int:
        return load().mb_viterbi_path_bound(self.h, inLen, outLen)


class DeviceBatch:
    """Device-resident tokenised pair list (mb_batch*)."""

    def __init__(self, dm: DeviceMachine, inTok, inOff, outTok, outOff):
        self.dm = dm
        self.inTok = np.ascontiguousarray(inTok, np.int32); self.outTok = np.ascontiguousarray(outTok, np.int32)
        self.inOff = np.ascontiguousarray(inOff, np.int64); self.outOff = np.ascontiguousarray(outOff, np.int64)
        self.nPairs = len(self.inOff) - 1
        assert len(self.outOff) == self.nPairs + 1
        L = load()
        self.h = L.mb_batch_create(dm.h, self.nPairs, _p(self.inTok, C.c_int32), _p(self.inOff, C.c_int64),
                                   _p(self.outTok, C.c_int32), _p(self.outOff, C.c_int64))
        if not self.h:
            raise MbError(L.mb_last_error().decode())

    @classmethod
    def from_pairs(cls, dm: DeviceMachine, pairs):
        """pairs: iterable of (inputTokens, outputTokens)."""
        pairs = list(pairs)
        inOff = np.zeros(len(pairs) + 1, np.int64); outOff = np.zeros(len(pairs) + 1, np.int64)
        for k, (a, b) in enumerate(pairs):
            inOff[k + 1] = inOff[k] + len(a); outOff[k + 1] = outOff[k] + len(b)
        cat = lambda xs: (np.concatenate([np.asarray(x, np.int32) for x in xs]) if len(xs) else np.zeros(0, np.int32)).astype(np.int32)
        return cls(dm, cat([a for a, _ in pairs]), inOff, cat([b for _, b in pairs]), outOff)

    def set_envelopes(self, envs):
        """envs: per pair either None (full) or (inStart, inEnd) arrays of outLen+1 entries (src/seqpair.h:75-97)."""
        off = np.zeros(self.nPairs + 1, np.int64)
        st, en = [], []
        for k, e in enumerate(envs):
            n = 0
            if e is not None:
                st.append(np.asarray(e[0], np.int32)); en.append(np.asarray(e[1], np.int32)); n = len(st[-1])
            off[k + 1] = off[k] + n
        cat = lambda xs: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(1, np.int32), np.int32)
        a, b = cat(st), cat(en)
        _check(load().mb_batch_set_envelopes(self.h, _p(off, C.c_int64), _p(a, C.c_int32), _p(b, C.c_int32)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            try:
                _lib.mb_batch_destroy(self.h)
            except Exception:
                pass
            self.h = None

    __del__ = close

    def cells(self) -> int:
        return load().mb_batch_cells(self.h)

    def forward(self, flags: int = MB_MATERIALISE) -> np.ndarray:
        ll = np.empty(self.nPairs, np.float64)
        _check(load().mb_batch_forward(self.h, flags, _p(ll, C.c_double)))
        return ll

    def viterbi(self, paths: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray], Optional[np.ndarray]]:
        ll = np.empty(self.nPairs, np.float64)
        if not paths:
            _check(load().mb_batch_viterbi(self.h, _p(ll, C.c_double), None, None, 0))
            return ll, None, None
        # mb_viterbi_path_bound is linear in the lengths: (inLen + outLen + 1) * levels + 1 per pair
        per = self.dm.path_bound(1, 0) - self.dm.path_bound(0, 0)
        lens = np.diff(self.inOff) + np.diff(self.outOff)
        cap = int((lens + 1).sum() * per + self.nPairs) if self.nPairs else 0
        off = np.zeros(self.nPairs + 1, np.int64); edges = np.empty(max(cap, 1), np.uint32)   # worst case; only the used part is touched
        _check(load().mb_batch_viterbi(self.h, _p(ll, C.c_double), _p(off, C.c_int64), _p(edges, C.c_uint32), cap))
        return ll, off, edges[:off[-1]]

    def counts(self, counts: Optional[np.ndarray] = None):
        """Returns (counts[nTrans], loglikeSum, loglike[nPairs]); accumulates into ``counts`` if given."""
        if counts is None:
            counts = np.zeros(self.dm.nTrans, np.float64)
        assert counts.dtype == np.float64 and counts.shape == (self.dm.nTrans,) and counts.flags.c_contiguous
        s = C.c_double(0.0)
        ll = np.empty(self.nPairs, np.float64)
        _check(load().mb_batch_counts(self.h, _p(counts, C.c_double), C.byref(s), _p(ll, C.c_double)))
        return counts, s.value, ll

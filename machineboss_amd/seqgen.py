"""Deterministic synthetic sequence pairs.

Symbols are i.i.d. uniform over the machine's alphabet from ``std::mt19937(seed)`` via
``alphabet[rng() % alphabet.size()]``; the input sequence is drawn before the output sequence.
Tokens are 1-based indices into the sorted alphabet (token 0 = epsilon, src/eval.h:17-22).
"""
from __future__ import annotations

import numpy as np


def mt19937_u32(seed: int, n: int) -> np.ndarray:
    """First ``n`` 32-bit outputs of std::mt19937(seed) (init_genrand seeding)."""
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    return rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)


def synth_tokens(seed: int, inLen: int, outLen: int, nInTok: int, nOutTok: int):
    r = mt19937_u32(seed, inLen + outLen)
    inp = (1 + (r[:inLen] % max(nInTok, 1))).astype(np.int32) if nInTok > 0 else np.zeros(0, np.int32)
    out = (1 + (r[inLen:inLen + outLen] % max(nOutTok, 1))).astype(np.int32) if nOutTok > 0 else np.zeros(0, np.int32)
    if nInTok == 0:
        out = (1 + (mt19937_u32(seed, outLen) % nOutTok)).astype(np.int32)
    return inp, out


def synth_batch(config: int, nPairs: int, inLen: int, outLen: int, nInTok: int, nOutTok: int, first: int = 0):
    """Pair k uses seed 1000*config + k.  Returns ragged CSR-style (tokens, offsets) for both tapes."""
    ins, outs = [], []
    for k in range(first, first + nPairs):
        a, b = synth_tokens(1000 * config + k, inLen, outLen, nInTok, nOutTok)
        ins.append(a); outs.append(b)
    inOff = np.zeros(nPairs + 1, np.int64); outOff = np.zeros(nPairs + 1, np.int64)
    inOff[1:] = np.cumsum([len(a) for a in ins]); outOff[1:] = np.cumsum([len(b) for b in outs])
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs and sum(len(x) for x in xs) else np.zeros(0, np.int32)
    return cat(ins), inOff, cat(outs), outOff

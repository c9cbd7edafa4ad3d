"""Random advancing transducers for parity tests (edge cases the presets do not have: match edges on medium
machines, duplicate edges (quirk Q10), several edges per (state, label), uneven silent levels, -inf weights)."""
import numpy as np

from machineboss_amd.evalmachine import EvaluatedMachine, Tokenizer


def random_machine(S, nIn, nOut, seed, density=2.0, silent_density=1.2, dup=True, allow_inf=False):
    rng = np.random.RandomState(seed)
    edges = []  # (src, dst, it, ot, logw)
    for s in range(S):
        n = rng.poisson(density) + 1
        for _ in range(n):
            kind = rng.randint(0, 3)
            it = rng.randint(1, nIn + 1) if kind in (0, 1) and nIn else 0
            ot = rng.randint(1, nOut + 1) if kind in (0, 2) and nOut else 0
            if it == 0 and ot == 0:
                continue
            edges.append((s, rng.randint(0, S), it, ot, float(np.log(rng.uniform(0.05, 1.0)))))
        ns = rng.poisson(silent_density)
        for _ in range(ns):
            if s + 1 < S:
                edges.append((s, rng.randint(s + 1, S), 0, 0, float(np.log(rng.uniform(0.05, 1.0)))))
        if s + 1 < S and rng.rand() < 0.7:   # keep the end state reachable through a silent backbone
            edges.append((s, s + 1, 0, 0, float(np.log(rng.uniform(0.2, 1.0)))))
    if dup and edges:
        for _ in range(max(1, len(edges) // 10)):
            e = edges[rng.randint(len(edges))]
            edges.append((e[0], e[1], e[2], e[3], float(np.log(rng.uniform(0.05, 1.0)))))
    if allow_inf and edges:
        k = rng.randint(len(edges)); e = edges[k]; edges[k] = (e[0], e[1], e[2], e[3], -np.inf)
    edges.sort(key=lambda e: e[0])   # stable: keeps insertion order within a source state
    n = len(edges)
    src = np.array([e[0] for e in edges], np.uint32); dst = np.array([e[1] for e in edges], np.uint32)
    it = np.array([e[2] for e in edges], np.uint16); ot = np.array([e[3] for e in edges], np.uint16)
    lw = np.array([e[4] for e in edges], np.float64)
    off = np.zeros(S + 1, np.int64)
    for s in src:
        off[s + 1] += 1
    off = np.cumsum(off)
    tidx = (np.arange(n) - off[src]).astype(np.uint32)
    return EvaluatedMachine(S, Tokenizer([chr(65 + k) for k in range(nIn)]), Tokenizer([chr(97 + k) for k in range(nOut)]),
                            src, dst, it, ot, tidx, lw, off, [None] * S)


def random_block_machine(blocks, per, nIn, nOut, seed, density=2.0, silent_density=1.2, allow_inf=False):
    """A one- or two-tape machine of `blocks` consecutive blocks of `per` states: an emitting transition ends in its source's own block
    (any state of it: cycles) or in a later block, a silent one further on in the state order -- so no cycle crosses a block boundary and
    the transition graph has cuts (the k-workgroups-per-sequence form of the one-tape family)."""
    rng = np.random.RandomState(seed)
    S = blocks * per
    edges = []
    for s in range(S):
        b0 = (s // per) * per
        for _ in range(rng.poisson(density) + 1):
            kind = rng.randint(0, 3)
            it = rng.randint(1, nIn + 1) if kind in (0, 1) and nIn else 0
            ot = rng.randint(1, nOut + 1) if kind in (0, 2) and nOut else 0
            if it == 0 and ot == 0: continue
            d = rng.randint(b0, min(S, b0 + per)) if rng.rand() < 0.7 else rng.randint(b0, S)
            edges.append((s, d, it, ot, float(np.log(rng.uniform(0.05, 1.0)))))
        for _ in range(rng.poisson(silent_density)):
            if s + 1 < S: edges.append((s, rng.randint(s + 1, min(S, s + 1 + 2 * per)), 0, 0, float(np.log(rng.uniform(0.05, 1.0)))))
        if s + 1 < S and rng.rand() < 0.7: edges.append((s, s + 1, 0, 0, float(np.log(rng.uniform(0.2, 1.0)))))
    if allow_inf and edges:
        k = rng.randint(len(edges)); e = edges[k]; edges[k] = (e[0], e[1], e[2], e[3], -np.inf)
    edges.sort(key=lambda e: e[0])
    n = len(edges)
    src = np.array([e[0] for e in edges], np.uint32); dst = np.array([e[1] for e in edges], np.uint32)
    it = np.array([e[2] for e in edges], np.uint16); ot = np.array([e[3] for e in edges], np.uint16)
    lw = np.array([e[4] for e in edges], np.float64)
    off = np.zeros(S + 1, np.int64)
    for q in src: off[q + 1] += 1
    off = np.cumsum(off)
    tidx = (np.arange(n) - off[src]).astype(np.uint32)
    return EvaluatedMachine(S, Tokenizer([chr(65 + k) for k in range(nIn)]), Tokenizer([chr(97 + k) for k in range(nOut)]),
                            src, dst, it, ot, tidx, lw, off, [None] * S)


def random_seq(rng, n, k):
    return rng.randint(1, k + 1, size=n).astype(np.int32) if k else np.zeros(0, np.int32)


def plain_hmm(S, fan, nOut, seed):
    """A generator in which every transition emits (a plain HMM: no silent level at all), `fan` transitions out of every
    state but the last -- what gives the one-tape family's retimed sweep a period of ONE stage."""
    rng = np.random.RandomState(seed)
    edges = [(s, rng.randint(1, S), rng.randint(1, nOut + 1), float(np.log(rng.uniform(0.05, 1.0)))) for s in range(S - 1) for _ in range(fan)]
    n = len(edges)
    src = np.array([e[0] for e in edges], np.uint32); dst = np.array([e[1] for e in edges], np.uint32)
    it = np.zeros(n, np.uint16); ot = np.array([e[2] for e in edges], np.uint16); lw = np.array([e[3] for e in edges], np.float64)
    off = np.zeros(S + 1, np.int64)
    for s in src:
        off[s + 1] += 1
    off = np.cumsum(off)
    tidx = (np.arange(n) - off[src]).astype(np.uint32)
    return EvaluatedMachine(S, Tokenizer([]), Tokenizer([chr(97 + k) for k in range(nOut)]), src, dst, it, ot, tidx, lw, off, [None] * S)

"""Sharding of a batch of sequence pairs over ranks.

Pairs are independent units (the `for seqPair` loops of target/boss.cpp:796,826 and src/counts.cpp:40-42), so the
batch shards with no data-path collective; only `--train`/`--counts` exchange anything: one all-reduce (sum) of the
nTransitions posterior counts + 1 log-likelihood per EM iteration (MachineCounts::operator+=, src/counts.cpp:66-71).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: (first, count) of rank's share of n items."""
    base, rem = divmod(n, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def lpt_assign(cells: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of pairs (by DP cell count) to ranks, for ragged batches."""
    order = sorted(range(len(cells)), key=lambda k: -int(cells[k]))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(k); load[r] += int(cells[k])
    for r in range(world):
        out[r].sort()
    return out


def allreduce_counts(counts: np.ndarray, loglike: float, backend_device: str = "cpu"):
    """Sum the E-step sufficient statistics over ranks (RCCL on GPUs -- backend "nccl" -- or gloo on CPU).

    Returns (counts, loglike) reduced in place; a no-op outside torch.distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return counts, loglike
    buf = torch.empty(counts.shape[0] + 1, dtype=torch.float64, device=backend_device)
    buf[:-1] = torch.from_numpy(counts).to(backend_device)
    buf[-1] = loglike
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    host = buf.cpu().numpy()
    counts[:] = host[:-1]
    return counts, float(host[-1])

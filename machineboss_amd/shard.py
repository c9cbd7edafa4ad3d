"""Sharding of a batch of sequence pairs over ranks.

Pairs are independent units (the `for seqPair` loops of target/boss.cpp:796,826 and src/counts.cpp:40-42), so the
batch shards with no data-path collective; only `--train`/`--counts` exchange anything: one all-reduce (sum) of the
nTransitions posterior counts + 1 log-likelihood per EM iteration (MachineCounts::operator+=, src/counts.cpp:66-71).
"""
from __future__ import annotations

import os
import sys
from typing import List, Optional, Sequence, Tuple

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


class RankGroup:
    """The ranks of a sharded run: one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, as
    torch.distributed.run sets them).

    ONE HIP runtime per process: rendezvous and the host-side odds and ends (barrier, max of a clock, gathering results in
    order) go over torch.distributed's `gloo` backend on the CPU -- torch never touches the GPU -- and the ONE data-path
    collective of the path, the sum of the E-step statistics (MachineCounts::operator+=, src/counts.cpp:66-71), goes over RCCL
    through the C-ABI (mb_comm_unique_id -> broadcast of the 128 bytes -> mb_comm_init -> mb_allreduce_counts) on the
    library's own runtime and stream, exactly as a C++ host would drive it.  Round 3 reduced through torch's
    RCCL: two HIP runtimes in one process (PyTorch bundles its own), whose load order decided which RCCL could see the device.

    backend: "rccl" (default on a GPU box) as above; "nccl": torch.distributed's NCCL backend (= torch's RCCL on torch's HIP
    runtime: the round-3 route, kept for comparison); "gloo": everything on the host (CPU tests, several ranks on one GPU).
    MB_DIST_BACKEND overrides the default."""

    def __init__(self, dist, rank: int, world: int, local_rank: int, backend: str, comm=None, opened: bool = False):
        self.dist, self.rank, self.world, self.local_rank, self.backend, self.comm, self._opened = dist, rank, world, local_rank, backend, comm, opened

    @classmethod
    def from_env(cls, backend: Optional[str] = None, force: bool = False, share_device: bool = False) -> Optional["RankGroup"]:
        """None outside a multi-rank launch (force = True: a one-rank group all the same -- the bootstrap and the collective
        exercised on a one-GPU box).  Binds the library to the rank's GPU BEFORE any GPU call."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 and not force:
            return None
        rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        from . import capi
        have_gpu = capi.device_count() > 0
        backend = backend or os.environ.get("MB_DIST_BACKEND") or ("rccl" if have_gpu else "gloo")
        if share_device:
            local = 0
        elif have_gpu and backend == "gloo" and local >= capi.device_count():
            local %= capi.device_count()      # all-host collectives: several ranks may share a GPU (a one-GPU box under torchrun --nproc-per-node 2)
        if have_gpu:
            # ranks that SHARE a device (the gloo dry runs on a one-GPU box; a host that co-locates processes) must not each claim 80 % of
            # its memory, and must not count on its CUs for themselves: the pools take a share (MB_MEM_FRACTION, read by the library when
            # it sizes a pool) and one-tape sweeps stay at one workgroup per sequence (k workgroups per sequence need their parts co-resident)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            ndev = max(capi.device_count(), 1)
            sharers = local_world if share_device else -(-local_world // ndev)
            if sharers > 1:
                os.environ.setdefault("MB_MEM_FRACTION", "%.4f" % (0.8 / sharers))
                os.environ.setdefault("MB_ONETAPE_PARTS", "1")
            capi.set_device(local)
        import torch.distributed as dist
        opened = False
        if not dist.is_initialized():
            if backend == "nccl":
                import torch
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            opened = True
        grp = cls(dist, rank, world, local, backend, None, opened)
        if backend == "rccl":
            import torch
            ident = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                ident = torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8).clone()
            if world > 1:
                dist.broadcast(ident, src=0)
            try:
                grp.comm = capi.Comm(bytes(ident.numpy().tobytes()), world, rank)      # bounded wait inside (mb_comm_init, MB_COMM_TIMEOUT_S)
            except Exception as e:
                # first contact gone wrong: say which rank and LEAVE -- the launcher (torch.distributed.run / bench.py's parent) reports the
                # failing rank and ends the others; a process that has touched the GPU is never re-exec'd or retried
                sys.stderr.write("[machineboss_amd] rank %d of %d (local rank %d): RCCL bootstrap failed: %s\n" % (rank, world, local, e))
                sys.stderr.flush()
                os._exit(17)
        return grp

    # ---- the data-path collective -----------------------------------------------------------------------------------------
    def allreduce_counts(self, counts: np.ndarray, loglike: float):
        """Sum of (posterior counts, log-likelihood) over the ranks, in place."""
        if self.comm is not None:
            return self.comm.allreduce_counts(counts, loglike)
        if self.world == 1:
            return counts, loglike
        return allreduce_counts(counts, loglike, "cuda" if self.backend == "nccl" else "cpu")

    # ---- host-side helpers (never on the data path) ------------------------------------------------------------------------
    def _dev(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def all_reduce_float(self, x: float, op: str = "sum") -> float:
        if self.world == 1:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_floats(self, row: Sequence[float]) -> List[List[float]]:
        if self.world == 1:
            return [[float(v) for v in row]]
        import torch
        rec = torch.tensor([float(v) for v in row], dtype=torch.float64, device=self._dev())
        out = [torch.zeros_like(rec) for _ in range(self.world)]
        self.dist.all_gather(out, rec)
        return [[float(v) for v in r.tolist()] for r in out]

    def close(self):
        if self.comm is not None:
            self.comm.close(); self.comm = None
        if self._opened and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self._opened = False

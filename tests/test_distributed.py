"""CPU tests of the multi-GPU plumbing (world_size 2, gloo): sharding is a partition, and the one exchange step of the
path -- the all-reduce of E-step sufficient statistics (MachineCounts::operator+=, src/counts.cpp:66-71) -- gives every
rank the serial result.  The per-rank E-step here is the oracle (no GPU in this container); on GPUs the same
allreduce_counts() runs over RCCL (backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_path
from machineboss_amd.shard import allreduce_counts, lpt_assign, shard_range


def test_shard_range_is_partition():
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                f, c = shard_range(n, w, r)
                got += list(range(f, f + c))
            assert got == list(range(n))


def test_lpt_assign_balances_ragged_batches():
    rng = np.random.RandomState(0)
    cells = rng.randint(1, 1000, size=200) ** 2
    parts = lpt_assign(cells, 8)
    assert sorted(k for p in parts for k in p) == list(range(200))
    loads = [sum(int(cells[k]) for k in p) for p in parts]
    assert max(loads) <= 1.05 * (sum(loads) / 8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_tokens
    from oracle import oracle
    m = Machine.fromFile(golden_path("preset", "dnapsw.json"))
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    om = oracle.OracleMachine(em)
    nPairs = 7
    first, count = shard_range(nPairs, world, rank)
    counts = np.zeros(em.nTransitions); ll = 0.0
    for k in range(first, first + count):
        x, y = synth_tokens(3000 + k, 20 + k, 25, em.nInTok, em.nOutTok)
        ll += om.counts_add(x, y, counts)
    counts, ll = allreduce_counts(counts, ll, "cpu")
    q.put((rank, counts.copy(), ll))
    dist.destroy_process_group()


def test_counts_allreduce_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    # serial reference
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_tokens
    from oracle import oracle
    em = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "dnapsw.json")), None, useDefaults=True)
    om = oracle.OracleMachine(em)
    ref = np.zeros(em.nTransitions); ref_ll = 0.0
    for k in range(7):
        x, y = synth_tokens(3000 + k, 20 + k, 25, em.nInTok, em.nOutTok)
        ref_ll += om.counts_add(x, y, ref)
    for rank, counts, ll in res:
        assert np.allclose(counts, ref, rtol=1e-12, atol=1e-14) and abs(ll - ref_ll) <= 1e-12 * abs(ref_ll)


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from machineboss_amd.boss import _gather_in_order, _shard
    from machineboss_amd.machine import Machine
    from machineboss_amd.seqpair import SeqPair
    machine = Machine.fromFile(golden_path("preset", "dnapsw.json"))
    # a ragged list: one long pair and six short ones (the CLI's shard: sorted by DP cell count, dealt greedily -- SURVEY 8(e))
    lens = [(10, 12), (400, 390), (11, 9), (30, 28), (8, 8), (25, 31), (12, 14)]
    data = [SeqPair(["A"] * a, ["C"] * b, "in%d" % k, "out%d" % k) for k, (a, b) in enumerate(lens)]
    owned = _shard(data, machine, world)
    mine = ["pair%d" % k for k in owned[rank]]
    q.put((rank, owned, _gather_in_order(mine, len(data), rank, world, owned)))
    dist.destroy_process_group()


def test_cli_results_gathered_in_input_order_world2():
    """`boss --loglike/--viterbi/--align` on N ranks: pairs are dealt by longest-processing-time-first on their DP cell
    counts (target/boss.cpp:796,826 loops over independent pairs), every rank computes the same assignment, results come
    back in input order (no data-path collective; a host-side object gather)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == res[1][1]                                  # the same partition on both ranks
    owned = res[0][1]
    assert sorted(k for part in owned for k in part) == list(range(7))
    assert [1] in owned                                            # the long pair alone on one rank, the six short ones on the other
    for rank, _, got in res:
        assert got == ["pair%d" % k for k in range(7)]


def _group_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from machineboss_amd.shard import RankGroup
    grp = RankGroup.from_env()                       # no GPU here: everything over gloo (on a GPU box: RCCL through the C-ABI)
    counts = np.arange(5, dtype=np.float64) * (rank + 1)
    counts, ll = grp.allreduce_counts(counts, -1.5 * (rank + 1))
    grp.barrier()
    q.put((rank, grp.backend, counts.copy(), ll, grp.all_reduce_float(float(rank + 1), "max"), grp.all_reduce_float(2.0, "sum"),
           grp.all_gather_floats([rank, 10.0 * rank])))
    grp.close()


def test_rank_group_world2():
    """shard.RankGroup, the ranks of bench.py and boss.py: environment as torch.distributed.run sets it, rendezvous over gloo,
    the count reduction and the host-side helpers (max of a clock, gather of per-rank rows) -- two processes, no GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, backend, counts, ll, mx, sm, rows in res:
        assert backend == "gloo" and np.array_equal(counts, np.arange(5) * 3.0) and ll == -4.5 and mx == 2.0 and sm == 4.0
        assert sorted(rows) == [[0.0, 0.0], [1.0, 10.0]]

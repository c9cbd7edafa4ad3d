"""Host overhead of the C-ABI around its kernels.

Round 5 shipped a 95 x wall-clock regression nobody saw: a materialised Forward that alternated with one-tape sweeps spent 7 s per
call in hipFree + hipMalloc of its 230 GB pool around 75 ms of kernels -- the budget that sizes the pool followed `free + cached`,
which moves by megabytes from call to call (a batch's tokens, a released block), and a pool is re-allocated when a request grows by
ANY amount.  The budget is sticky now (mb_api.hip budget_bytes) and `mb_alloc_stats` makes the pools' allocations visible.  These
tests alternate the two kinds of sweep the way bench.py's `extra` blocks do, with the device's free memory perturbed between the
calls, and assert that the later calls allocate nothing and that their wall time is their device time.
MB_MEM_FRACTION shrinks the budget so that the pools are gigabytes, not hundreds of gigabytes (a 230 GB hipMalloc is 5-7 s)."""
import time

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from machineboss_amd import capi as c
    if c.device_count() == 0:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return c


def _profile_machine(nodes):
    from machineboss_amd import algebra as A
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
    h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(nodes)
    m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    return EvaluatedMachine.fromMachine(m, None, useDefaults=True)


def test_alternating_sweeps_do_not_reallocate_the_pools(capi, monkeypatch):
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_batch
    monkeypatch.setenv("MB_MEM_FRACTION", "0.02")            # ~5 GB of budget on a 288 GB device
    monkeypatch.setenv("MB_ONETAPE_PARTS_MIN_LEN", "0")      # the one-tape sweeps run k workgroups per sequence (exchange buffers in slots 13 / 14)
    monkeypatch.setenv("MB_ROLLING_MIN_PAIRS", "192")
    capi.release_workspace()
    em4 = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "psw2dna.json")), None, useDefaults=True)
    dm4 = capi.DeviceMachine(em4)
    # 12 pairs of 487 x 2000: 12 x 2.1 GB of fp64 cells against ~5 GB of budget -- the pipeline over recycled matrix slots
    b4 = capi.DeviceBatch(dm4, *synth_batch(4, 12, 487, 2000, em4.nInTok, em4.nOutTok))
    em5 = _profile_machine(3)
    dm5 = capi.DeviceMachine(em5)
    b5 = capi.DeviceBatch(dm5, *synth_batch(5, 8, 0, 3000, em5.nInTok, em5.nOutTok))
    ref4 = ref5 = None
    rows = []
    for rep in range(5):
        # what moved the budget in round 5: device memory that comes and goes between the calls -- here a batch of tokens that is alive
        # during the materialised Forward and SMALLER in every repetition, so `free + cached` grows from call to call
        junk = capi.DeviceBatch(dm4, *synth_batch(4, 2, 300, 20000000 // (rep + 1), em4.nInTok, em4.nOutTok)) if rep < 4 else None
        a0 = capi.alloc_stats()
        t0 = time.perf_counter(); ll5 = b5.forward(capi.MB_ROLLING); w5 = time.perf_counter() - t0; d5 = capi.last_device_ms(); k5 = capi.last_kernel_name()
        t0 = time.perf_counter(); v5 = b5.viterbi(paths=False); w5v = time.perf_counter() - t0; d5v = capi.last_device_ms()
        t0 = time.perf_counter(); ll4 = b4.forward(capi.MB_MATERIALISE); w4 = time.perf_counter() - t0; d4 = capi.last_device_ms()
        a1 = capi.alloc_stats()
        del junk
        rows.append((rep, w5 * 1e3, d5, w5v * 1e3, d5v, w4 * 1e3, d4, a1["pool_allocs"] - a0["pool_allocs"], a1["pool_frees"] - a0["pool_frees"], a1["ms"] - a0["ms"]))
        if rep == 0:
            ref4, ref5 = ll4, ll5
            assert "parts" in k5, k5
        else:
            assert np.array_equal(ll4, ref4) and np.allclose(ll5, ref5, rtol=1e-9)
    print(rows)
    for rep, w5, d5, w5v, d5v, w4, d4, na, nf, ms in rows[2:]:      # (rep 0 builds programs and pools, rep 1 may still settle the budget downwards)
        assert na == 0 and nf == 0, rows
        assert w4 <= 1.2 * d4 + 5.0, rows       # ms: wall against HIP-event time of the sweeps
        assert w5 <= 1.2 * d5 + 5.0 and w5v <= 1.2 * d5v + 5.0, rows


def test_alloc_stats_count_a_forced_reallocation(capi, monkeypatch):
    """mb_alloc_stats sees what the pools cost: releasing the workspace and calling again allocates again."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_batch
    em = EvaluatedMachine.fromMachine(Machine.fromFile(golden_path("preset", "psw2dna.json")), None, useDefaults=True)
    dm = capi.DeviceMachine(em)
    b = capi.DeviceBatch(dm, *synth_batch(4, 2, 100, 400, em.nInTok, em.nOutTok))
    b.forward(capi.MB_MATERIALISE)
    a0 = capi.alloc_stats()
    b.forward(capi.MB_MATERIALISE)
    a1 = capi.alloc_stats()
    assert a1["pool_allocs"] == a0["pool_allocs"] and a1["pool_frees"] == a0["pool_frees"]
    capi.release_workspace()
    b.forward(capi.MB_MATERIALISE)
    a2 = capi.alloc_stats()
    assert a2["pool_allocs"] > a1["pool_allocs"] and a2["pool_frees"] > a1["pool_frees"] and a2["bytes_allocated"] > a1["bytes_allocated"]

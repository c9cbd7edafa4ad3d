"""The drop-in classes as the reference's OWN call sites drive them: tests/cxx/dropin.cpp runs the loops of
target/boss.cpp:796-800 (`--loglike`) and :826-833 (`--viterbi / --align`) as written there -- one matrix object per pair --
through machineboss_amd/cxx/mb_dp.hpp.

* no fp64 matrix crosses PCIe for logLike() / path(machine) (`matrixFills()` stays 0 over both loops);
* with the ONE added line (`MachineBossHIP::prefetch`) the loops return the same numbers and paths from one batched call;
* `cell()` AFTER a lazy construction still equals the oracle (Viterbi bit for bit, Forward within the fast-path tolerance),
  `path(machine)` equals the oracle's traceBack, and the host walker over the lazily fetched matrix agrees with the device's path.
dnapsw (small family), psw2dna (tiled family), a one-tape profile composite (one-tape family, traceback codes).
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_path

sys.path.insert(0, os.path.join(ROOT, "tests", "cxx"))
import casefile  # noqa: E402

FAST_REL, FAST_ABS = 2e-6, 2e-5


def test_dropin_harness_compiles(tmp_path):
    from machineboss_amd import build
    build.build()
    assert os.path.exists(casefile.build_exe(tmp_path, "dropin"))


def _case(name):
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_tokens
    if name == "onetape":
        from machineboss_amd import algebra as A
        from machineboss_amd.hmmer import HmmerModel
        P = lambda n: Machine.fromFile(golden_path("preset", n + ".json"))
        h = HmmerModel.fromFile(golden_path("hmmer", "fn3.hmm")).truncated(3)
        m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
        shapes = [(0, 37), (0, 70), (0, 1)]
    else:
        m = Machine.fromFile(golden_path("preset", name + ".json"))
        shapes = [(45, 52), (3, 80), (70, 9)] if name == "dnapsw" else [(9, 31), (14, 20)]
    em = casefile.file_weights(EvaluatedMachine.fromMachine(m, None, useDefaults=True))
    pairs = [synth_tokens(90 + k, il, ol, em.nInTok, em.nOutTok) for k, (il, ol) in enumerate(shapes)]
    return em, pairs


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dnapsw", "psw2dna", "onetape"])
def test_reference_call_sites_through_the_lazy_classes(tmp_path, oracle_mod, name):
    from machineboss_amd import capi
    if capi.device_count() == 0:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    em, pairs = _case(name)
    case = str(tmp_path / "case.txt")
    casefile.write_case(case, em, ["s%d" % s for s in range(em.nStates)], pairs)
    exe = casefile.build_exe(tmp_path, "dropin")
    env = dict(os.environ); env.pop("MB_ROLLING_MIN_PAIRS", None)
    out = subprocess.run([exe, case, "check"], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "DROPIN OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    om = oracle_mod.OracleMachine(em)
    sym_in, sym_out = em.inputTokenizer.tok2sym, em.outputTokenizer.tok2sym
    as_text = lambda edges: ["%d,%s,%s" % (em.dst[e], sym_in[em.inTok[e]] if em.inTok[e] else "-", sym_out[em.outTok[e]] if em.outTok[e] else "-") for e in edges]
    i_loop, i_pre = lines.index("LOOP"), lines.index("PREFETCH")
    sections = {"loop": lines[i_loop + 1:i_pre], "prefetch": lines[i_pre + 1:]}
    ref_ll = [om.loglike(x, y, oracle_mod.SUM_EXACT) for x, y in pairs]
    Vs = [om.viterbi(x, y) for x, y in pairs]
    for tag, sec in sections.items():
        ll = [float(l.split()[1]) for l in sec if l.startswith("loglike ")][:len(pairs)]
        vit = [float(l.split()[1]) for l in sec if l.startswith("viterbi ")][:len(pairs)]
        al = [l.split()[1:] for l in sec if l.startswith("align")][:len(pairs)]
        assert len(ll) == len(pairs) and len(vit) == len(pairs)
        ai = 0
        for k, (x, y) in enumerate(pairs):
            assert abs(ll[k] - ref_ll[k]) <= FAST_REL * abs(ref_ll[k]) + FAST_ABS if math.isfinite(ref_ll[k]) else ll[k] == ref_ll[k], (tag, k)
            assert vit[k] == Vs[k][-1, -1, -1], (tag, k)                       # bit for bit
            if Vs[k][-1, -1, -1] > -math.inf:
                assert al[ai] == as_text(om.traceback(x, y, Vs[k])), (tag, k)   # the reference's first-maximum path
                ai += 1
    # logLike() and path(machine) of both loops, with and without the prefetch, moved no matrix
    assert [l for l in lines if l.startswith("fills_after_loops")][0].split()[1] == "0"
    assert [l for l in lines if l.startswith("fills_after_prefetch_loops")][0].split()[1] == "0"
    # cell() after the lazy construction (from prefetched results): the matrices are fetched then, and are the oracle's
    lazy = [l.split() for l in lines if l.startswith("lazy ")]
    vc = [l.split() for l in lines if l.startswith("vcells ")]
    fc = [l.split() for l in lines if l.startswith("fcells ")]
    fp = [l.split() for l in lines if l.startswith("fills_for_pair ")]
    assert len(lazy) == len(pairs)
    for k, (x, y) in enumerate(pairs):
        assert lazy[k][-1] == "00"                                               # nothing fetched by construction + logLike()
        assert float(lazy[k][2]) == Vs[k][-1, -1, -1]
        V = np.array([float(t) for t in vc[k][2:]]).reshape(Vs[k].shape)
        assert np.array_equal(V, Vs[k])
        F = np.array([float(t) for t in fc[k][2:]]).reshape(Vs[k].shape)
        Fo = om.forward(x, y, oracle_mod.SUM_EXACT)
        fin = np.isfinite(Fo)
        assert np.array_equal(np.isfinite(F), fin) and np.all(np.abs(F[fin] - Fo[fin]) <= FAST_REL * np.abs(Fo[fin]) + FAST_ABS)
        assert fp[k][1] == "2" and float(fp[k][3]) == Vs[k][-1, -1, -1]          # one fetch per matrix, and endCell() == logLike()
        assert abs(float(fp[k][4]) - float(lazy[k][3])) <= 1e-8 * abs(float(lazy[k][3])) + 1e-12 if math.isfinite(float(lazy[k][3])) else True
    assert all(l.split()[1] == "1" for l in lines if l.startswith("walker_agrees"))
    api = [l.split() for l in lines if l.startswith("api ")][0]
    assert float(api[2]) == Vs[0][-1, -1, -1] and int(api[3]) == len(om.traceback(pairs[0][0], pairs[0][1], Vs[0]))

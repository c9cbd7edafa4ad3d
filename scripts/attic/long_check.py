import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
for preset, n, il, ol in (("psw2dna", 2, 487, 50000), ("dnapsw", 2, 20000, 20000), ("protpsw", 3, 5000, 3000)):
    m = Machine.fromFile("/root/repo/tests/golden/preset/%s.json" % preset); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    dm = capi.DeviceMachine(em); b = capi.DeviceBatch(dm, *synth_batch(9, n, il, ol, em.nInTok, em.nOutTok))
    t = time.perf_counter(); llm = b.forward(capi.MB_MATERIALISE); t1 = time.perf_counter() - t
    llr = b.forward(capi.MB_ROLLING)
    v, off, edges = b.viterbi()
    e = edges[off[0]:off[1]]
    ok_path = em.src[e[0]] == 0 and em.dst[e[-1]] == em.nStates - 1 and np.array_equal(em.dst[e[:-1]], em.src[e[1:]])
    acc = 0.0
    for w in np.asarray(em.logWeight)[e]: acc += w
    print(preset, il, ol, "mat %.1f Gcells/s" % (b.cells() / t1 / 1e9), llm, np.max(np.abs(llm - llr) / np.abs(llm)), "viterbi", v[0], "path ok", ok_path, abs(acc - v[0]) < 1e-6 * abs(v[0]), len(e))
    capi.release_workspace()

import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
m = Machine.fromFile('/root/repo/tests/golden/preset/dnapsw.json')
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(2, 1024, 1000, 1000, em.nInTok, em.nOutTok))
for paths in (False, True, False, True):
    t0 = time.perf_counter(); r = b.viterbi(paths=paths); dt = time.perf_counter() - t0
    print('paths', paths, 'wall %.1f ms  device %.1f ms' % (dt * 1e3, capi.last_device_ms()), (len(r[2]) if paths else ''))

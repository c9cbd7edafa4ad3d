import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
mode = sys.argv[1] if len(sys.argv) > 1 else "rolling"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
outlen = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
m = Machine.fromFile("tests/golden/preset/psw2dna.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(4, pairs, 487, outlen, em.nInTok, em.nOutTok))
fl = capi.MB_ROLLING if mode == "rolling" else capi.MB_MATERIALISE
for r in range(reps):
    t = time.perf_counter(); ll = b.forward(fl); dt = time.perf_counter() - t
    print(mode, "rep", r, "%.1f Gcells/s" % (b.cells() / dt / 1e9), "dev ms %.2f" % capi.last_device_ms(), capi.last_kernel_name(), flush=True)

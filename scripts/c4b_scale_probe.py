"""The 482-state E-step (protpsw . translate . dnapsw) against the number of pairs in the call: is 24 x 487 aa x 10 kb bound by the
chip's throughput or by the pairs' own dependency chains?  usage: c4b_scale_probe.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
ns = [int(a) for a in sys.argv[1:]] or [3, 6, 12, 24]
em = EvaluatedMachine.fromMachine(algebra.config4bMachine("tests/golden/preset"), None, useDefaults=True)
dm = capi.DeviceMachine(em)
for n in ns:
    b = capi.DeviceBatch(dm, *synth_batch(4, n, 487, 10000, em.nInTok, 3))
    b.counts()
    t0 = time.perf_counter(); c, s, _ = b.counts(); dt = time.perf_counter() - t0
    lat = n * 488 * 10001 * em.nStates
    print("%3d pairs: %8.1f ms wall %8.1f ms device  %6.1f G lattice-cells/s  launches %d  %s" % (n, dt * 1e3, capi.last_device_ms(), lat / dt / 1e9, capi.last_launch_count(), capi.last_kernel_name()), flush=True)
    del b

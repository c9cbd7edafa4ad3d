"""psw2dna count sweep (tiled family, no Forward matrix) under tuning knobs: device ms of one mb_batch_counts call.
usage: python scripts/count_probe.py [nPairs inLen outLen]   (knobs from the environment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
n, il, ol = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 487, 2000)
m = Machine.fromFile("tests/golden/preset/psw2dna.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(4, n, il, ol, em.nInTok, em.nOutTok))
b.counts()
ms = []
for _ in range(3):
    c = b.counts(); ms.append(capi.last_device_ms())
knobs = {k: v for k, v in os.environ.items() if k.startswith("MB_")}
print("%d x %d x %d counts: device %.2f ms  %.1f G lattice-cells/s  launches %d  checksum %.6f  %s" % (n, il, ol, min(ms), b.cells() / min(ms) / 1e6, capi.last_launch_count(), float(c[0].sum()), knobs), flush=True)

import sys, time
sys.path.insert(0, "/root/repo")
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
preset, n, il, ol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = Machine.fromFile("/root/repo/tests/golden/preset/%s.json" % preset); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em); b = capi.DeviceBatch(dm, *synth_batch(2, n, il, ol, em.nInTok, em.nOutTok))
b.counts(); b.counts()

"""Accuracy of the posterior-count sweep: symbol-count invariant (sum of counts of input-consuming transitions = number
of input symbols) and agreement with the generic fp64 sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
preset, pairs, il, ol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = Machine.fromFile("tests/golden/preset/%s.json" % preset); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(2, pairs, il, ol, em.nInTok, em.nOutTok))
c, s, ll = b.counts()
cin = c[np.asarray(em.inTok) != 0].sum(); cout = c[np.asarray(em.outTok) != 0].sum()
print(capi.last_kernel_name(), "in-invariant rel err %.2e  out %.2e" % (abs(cin / (pairs * il) - 1), abs(cout / (pairs * ol) - 1)))
np.save("/tmp/counts_%s.npy" % os.environ.get("TAG", "x"), c)

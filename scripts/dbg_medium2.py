import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
from oracle import oracle
il, ol = int(sys.argv[1]), int(sys.argv[2])
m = Machine.fromFile("tests/golden/preset/psw2dna.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
om = oracle.OracleMachine(em)
x, y = synth_tokens(11, il, ol, em.nInTok, em.nOutTok)
refs = {capi.MB_VITERBI: om.viterbi(x, y), capi.MB_FORWARD: om.forward(x, y, oracle.SUM_EXACT), capi.MB_BACKWARD: om.backward(x, y, oracle.SUM_EXACT)}
for G in sys.argv[3:]:
    os.environ["MB_MEDIUM_G"] = G
    dm = capi.DeviceMachine(em)
    for mode, name in ((capi.MB_VITERBI, "vit"), (capi.MB_FORWARD, "fwd"), (capi.MB_BACKWARD, "bwd")):
        A = dm.fill(mode, x, y); R = refs[mode]
        fin = np.isfinite(R)
        bad = (np.isfinite(A) != fin) | (fin & (np.abs(A - np.where(fin, R, 0)) > 2e-5 + 2e-6 * np.abs(np.where(fin, R, 0))))
        msg = ""
        if bad.any():
            idx = np.argwhere(bad); o, i, s = idx[0]
            msg = " first bad (o,i,s)=%s got %r want %r; o range %d..%d i range %d..%d nanA=%d" % (idx[0].tolist(), A[o, i, s], R[o, i, s], idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(), int(np.isnan(A).sum()))
        print("G", G, name, "bad", int(bad.sum()), msg)
    dm.close()

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
from oracle import oracle
G = sys.argv[1] if len(sys.argv) > 1 else "4"
os.environ["MB_MEDIUM_G"] = G
il, ol = int(sys.argv[2]), int(sys.argv[3])
m = Machine.fromFile("tests/golden/preset/psw2dna.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
om = oracle.OracleMachine(em); dm = capi.DeviceMachine(em)
x, y = synth_tokens(11, il, ol, em.nInTok, em.nOutTok)
for mode, name in ((capi.MB_VITERBI, "vit"), (capi.MB_FORWARD, "fwd"), (capi.MB_BACKWARD, "bwd")):
    A = dm.fill(mode, x, y)
    R = om.viterbi(x, y) if mode == capi.MB_VITERBI else (om.forward(x, y, oracle.SUM_EXACT) if mode == capi.MB_FORWARD else om.backward(x, y, oracle.SUM_EXACT))
    fin = np.isfinite(R)
    bad = (np.isfinite(A) != fin) | (fin & (np.abs(A - np.where(fin, R, 0)) > 2e-5 + 2e-6 * np.abs(np.where(fin, R, 0))))
    print(name, capi.last_kernel_name(), "bad cells", int(bad.sum()), "of", bad.size)
    if bad.any():
        idx = np.argwhere(bad)
        print(" first bad (o,i,s):", idx[:5].tolist(), "min o", idx[:, 0].min(), "min i", idx[:, 1].min(), "i set", sorted(set(idx[:, 1].tolist()))[:20])
        o, i, s = idx[0]
        print(" got", A[o, i, s], "want", R[o, i, s])
b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
print("rolling", b.forward(capi.MB_ROLLING)[0], "mat", b.forward(capi.MB_MATERIALISE)[0], "oracle", om.loglike(x, y, oracle.SUM_EXACT))

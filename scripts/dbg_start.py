import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
from oracle import oracle
m = Machine.fromFile("tests/golden/preset/dnapsw.json"); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
om = oracle.OracleMachine(em); dm = capi.DeviceMachine(em)
x, y = synth_tokens(7, 37, 53, 4, 4)
for st in (0, 2):
    A = dm.fill(capi.MB_FORWARD, x, y, startState=st); R = om.forward(x, y, oracle.SUM_EXACT, startState=st)
    fin = np.isfinite(R)
    bad = (np.isfinite(A) != fin) | (fin & (np.abs(A - np.where(fin, R, 0)) > 2e-5 + 2e-6 * np.abs(np.where(fin, R, 0))))
    print("start", st, capi.last_kernel_name(), "bad", int(bad.sum()), "of", bad.size)
    if bad.any():
        idx = np.argwhere(bad); print(idx[:6].tolist()); o, i, s = idx[0]; print("got", A[o, i, s], "want", R[o, i, s]); print("origin got", A[0,0], "want", R[0,0])

"""Every mode of the hot path on one machine and batch shape: device ms and G cells/s per mode (knobs from the environment).
usage: python scripts/mode_probe.py <preset|c4b> nPairs inLen outLen [modes=fwd,roll,vit,cnt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
preset = sys.argv[1]; n, il, ol = (int(x) for x in sys.argv[2:5])
modes = (sys.argv[5] if len(sys.argv) > 5 else "fwd,roll,vit,cnt").split(",")
if preset == "c4b":
    from machineboss_amd import algebra
    m = algebra.config4bMachine("tests/golden/preset")
else:
    m = Machine.fromFile("tests/golden/preset/%s.json" % preset)
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(4, n, il, ol, em.nInTok, 3 if preset == "c4b" else em.nOutTok))   # c4b: DNA over {A,C,G}, no stop codons
fns = {"fwd": lambda: b.forward(capi.MB_MATERIALISE), "roll": lambda: b.forward(capi.MB_ROLLING), "vit": lambda: b.viterbi(paths=True), "cnt": lambda: b.counts()}
knobs = {k: v for k, v in os.environ.items() if k.startswith("MB_")}
out = []
for md in modes:
    r = fns[md](); ms = []
    for _ in range(3):
        r = fns[md](); ms.append(capi.last_device_ms())
    chk = float(np.sum(r[0])) if isinstance(r, tuple) else float(np.sum(r))
    out.append("%s %.2f ms %.1f G/s [%s] chk %.6f" % (md, min(ms), b.cells() / min(ms) / 1e6, capi.last_kernel_name(), chk))
print("%s %d x %d x %d: %s  %s" % (preset, n, il, ol, " | ".join(out), knobs), flush=True)

"""The C4b count sweep with PyTorch imported FIRST (libmbhip.so then binds to torch's bundled libamdhip64 / libhiprtc, ROCm 7.0,
instead of /opt/rocm's 7.2): posterior counts of one 5 x 18 pair against the oracle.  Knobs from the environment."""
import os, sys
if os.environ.get("PROBE_TORCH", "1") == "1":
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
from oracle import oracle
em = EvaluatedMachine.fromMachine(algebra.config4bMachine("tests/golden/preset"), None, useDefaults=True)
om = oracle.OracleMachine(em); dm = capi.DeviceMachine(em)
res = []
for il, ol in ((5, 18), (40, 300)):
    x, y = synth_tokens(12, il, ol, em.nInTok, 3)
    b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
    counts, s, cll = b.counts()
    ref = np.zeros(em.nTransitions); ll = om.counts_add(x, y, ref, oracle.SUM_EXACT)
    err = np.abs(counts - ref).max() / max(1.0, np.abs(ref).max())
    res.append("%dx%d ll %.6f vs %.6f err %.3g" % (il, ol, cll[0], ll, err))
maps = open("/proc/self/maps").read()
rt = "torch" if "torch/lib/libhiprtc" in maps else "system"
print(rt, "hiprtc |", " | ".join(res), "|", {k: v for k, v in os.environ.items() if k.startswith("MB_")}, flush=True)

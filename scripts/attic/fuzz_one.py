"""One case of scripts/fuzz_gpu.py by (seed0, case): prints the rolling / materialised log-likelihoods of both families."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from randmachine import random_machine, random_seq
seed0, c = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed0 + c)
oneTape = c % 3 == 2
S = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 16, 17, 33, 64, 100, 257, 300, 700])) if not oneTape else int(rng.choice([5, 40, 260, 300, 900]))
nIn = 0 if oneTape else int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
if oneTape and c % 6 == 5: nIn, nOut = nOut, 0
em = random_machine(S, nIn, nOut, seed0 + c, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=bool(c % 5 == 0))
if oneTape: os.environ["MB_WIDE_MIN_STATES"] = "1" if c % 2 else "256"
dm = capi.DeviceMachine(em)
n = int(rng.randint(1, 6))
scale = 6 if (c % 7 == 3 and S <= 100) or (c % 2 == 1 and S <= 16) else 1
lo = 64 if oneTape and c % 4 == 2 else 0
pairs = [(random_seq(rng, int(rng.randint(lo, lo + 40 * scale)) if nIn else 0, nIn), random_seq(rng, int(rng.randint(lo, lo + 60 * scale)) if nOut else 0, nOut)) for _ in range(n)]
b = capi.DeviceBatch.from_pairs(dm, pairs)
print("S", S, "nIn", nIn, "nOut", nOut, "pairs", [(len(x), len(y)) for x, y in pairs])
for rep in range(3):
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
        capi.set_kernel(fam)
        ll = b.forward(capi.MB_ROLLING); k1 = capi.last_kernel_name(); llm = b.forward(capi.MB_MATERIALISE); k2 = capi.last_kernel_name()
        print(rep, "family", fam, k1, ll, k2, llm)
    capi.set_kernel(capi.KERNEL_AUTO)

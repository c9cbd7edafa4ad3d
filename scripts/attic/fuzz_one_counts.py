"""One case of scripts/fuzz_gpu.py by (seed0, case): posterior counts of the automatic family, the generic family and the oracle."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from machineboss_amd import capi
from oracle import oracle
from randmachine import random_machine, random_seq
seed0, c = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed0 + c)
oneTape = c % 3 == 2
S = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 16, 17, 33, 64, 100, 257, 300, 700])) if not oneTape else int(rng.choice([5, 40, 260, 300, 900]))
nIn = 0 if oneTape else int(rng.randint(1, 4)); nOut = int(rng.randint(1, 4))
if oneTape and c % 6 == 5: nIn, nOut = nOut, 0
em = random_machine(S, nIn, nOut, seed0 + c, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=bool(c % 5 == 0))
if oneTape: os.environ["MB_WIDE_MIN_STATES"] = "1" if c % 2 else "256"
dm = capi.DeviceMachine(em); om = oracle.OracleMachine(em)
n = int(rng.randint(1, 6))
scale = 6 if (c % 7 == 3 and S <= 100) or (c % 2 == 1 and S <= 16) else 1
lo = 64 if oneTape and c % 4 == 2 else 0
pairs = [(random_seq(rng, int(rng.randint(lo, lo + 40 * scale)) if nIn else 0, nIn), random_seq(rng, int(rng.randint(lo, lo + 60 * scale)) if nOut else 0, nOut)) for _ in range(n)]
b = capi.DeviceBatch.from_pairs(dm, pairs)
ref = np.zeros(em.nTransitions); lls = []
for x, y in pairs:
    if om.loglike(x, y, oracle.SUM_EXACT) > -math.inf: lls.append(om.counts_add(x, y, ref, oracle.SUM_EXACT))
    else: lls.append(-math.inf)
print("S", S, "nIn", nIn, "nOut", nOut, "T", em.nTransitions, "pairs", [(len(x), len(y)) for x, y in pairs], "oracle ll", lls)
for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
    capi.set_kernel(fam)
    cnt, s, ll = b.counts()
    d = np.abs(cnt - ref); k = int(np.argmax(d / np.maximum(np.abs(ref), 1e-6)))
    print("family", fam, capi.last_kernel_name(), "ll", ll, "max abs dev %.3g at edge %d (ref %.6g got %.6g; src %d dst %d in %d out %d)" % (d.max(), k, ref[k], cnt[k], em.src[k], em.dst[k], em.inTok[k], em.outTok[k]),
          {kk: v for kk, v in os.environ.items() if kk.startswith("MB_")})
capi.set_kernel(capi.KERNEL_AUTO)
if os.environ.get("FUZZ_DETAIL"):
    capi.set_kernel(capi.KERNEL_AUTO)
    cnt, s, ll = b.counts()
    it = np.asarray(em.inTok); ot = np.asarray(em.outTok); src = np.asarray(em.src); dst = np.asarray(em.dst)
    bad = np.where(np.abs(cnt - ref) > 1e-4 * np.abs(ref) + 1e-6)[0]
    print("bad edges", len(bad))
    for T, name in ((0, "match"), (1, "in"), (2, "out"), (3, "sil")):
        for a in range(em.nInTok + 1):
            for o in range(em.nOutTok + 1):
                kind = 0 if (a and o) else (1 if a else (2 if o else 3))
                if kind != T: continue
                lst = [e for e in range(em.nTransitions) if it[e] == a and ot[e] == o and not (T == 3 and dst[e] <= src[e])]
                lst.sort(key=lambda e: dst[e])
                marks = "".join("X" if e in set(bad) else "." for e in lst)
                if lst: print(name, "tok", a, o, "n", len(lst), marks)
    print("dst of bad edges", sorted(set(int(dst[e]) for e in bad)), "src", sorted(set(int(src[e]) for e in bad))[:40])

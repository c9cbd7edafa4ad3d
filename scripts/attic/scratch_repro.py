"""Round-2 open item: one randomised case (scripts/fuzz_env_gpu.py seed 5229: 12 states, 69 transitions, enveloped batch) whose
count sweep, compiled for 3 wavefronts per SIMD, spills 469 VGPRs to scratch memory and returned wrong values, while the
same source compiled for 2 (268 spills) or 1 (no scratch) did not.  This script runs that case under a set of build
variants, each in its own process, and prints which outputs differ from the generic family -- to tell a compiler problem
(spill code) from a scratch-memory problem (runtime / hardware) from a latent bug of the kernel that only this build exposes.

usage: python scripts/scratch_repro.py [seed]          (parent: runs every variant)
       python scripts/scratch_repro.py [seed] child    (one variant, configuration taken from the environment)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))

VARIANTS = [
    ("default (no scratch allowed)", {}),
    ("3 waves/SIMD, 469 spills", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3"}),
    ("3 waves/SIMD, 469 spills, run again", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3"}),
    ("2 waves/SIMD, 268 spills", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "2"}),
    ("4 waves/SIMD", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "4"}),
    ("3 waves/SIMD, SGPRs spilled to memory instead of VGPR lanes", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3", "MB_JIT_EXTRA_OPTS": "-mllvm -amdgpu-spill-sgpr-to-vgpr=0"}),
    ("3 waves/SIMD, -O1", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3", "MB_JIT_EXTRA_OPTS": "-O1"}),
    ("3 waves/SIMD, every lane stores/loads (no line masks)", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3", "MB_SMALL_STORE_LINES": "0"}),
    ("3 waves/SIMD, Backward loads late", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3", "MB_SMALL_BLOAD": "1"}),
    ("3 waves/SIMD, 128-step tiles", {"MB_SMALL_ALLOW_SCRATCH": "1", "MB_SMALL_MINWAVES": "3", "MB_SMALL_TS": "128"}),
]


def child(seed):
    import numpy as np
    import fuzz_env_gpu as fz
    from machineboss_amd import capi
    em, pairs, envs = fz.make_case(seed)
    print("  machine S=%d T=%d, pairs %s, enveloped %s" % (em.nStates, em.nTransitions, [(len(a), len(b)) for a, b in pairs], [e is not None for e in envs]))
    out = {}
    for fam in (capi.KERNEL_AUTO, capi.KERNEL_GENERIC):
        capi.set_kernel(fam)
        dm = capi.DeviceMachine(em)
        b = capi.DeviceBatch.from_pairs(dm, pairs)
        b.set_envelopes([(e.inStart, e.inEnd) if e is not None else None for e in envs])
        rep = []
        for _ in range(3 if fam == capi.KERNEL_AUTO else 1):
            c, s, ll = b.counts()
            rep.append((np.array(c), float(s), np.array(ll)))
        out[fam] = rep + [capi.last_kernel_name()]
        # one pair at a time: is a wrong value tied to the batch (tile lists, neighbours) or to the pair?
        if fam == capi.KERNEL_AUTO:
            single = []
            for k, (x, y) in enumerate(pairs):
                b1 = capi.DeviceBatch.from_pairs(dm, [(x, y)])
                b1.set_envelopes([(envs[k].inStart, envs[k].inEnd) if envs[k] is not None else None])
                single.append(b1.counts())
            out["single"] = single
    capi.set_kernel(capi.KERNEL_AUTO)
    g = out[capi.KERNEL_GENERIC][0]
    print("  kernel", out[capi.KERNEL_AUTO][-1], "jit", capi.jit_stats() if hasattr(capi, "jit_stats") else "")
    for r, (c, s, ll) in enumerate(out[capi.KERNEL_AUTO][:-1]):
        dll = np.abs(ll - g[2]); dll[np.isneginf(ll) & np.isneginf(g[2])] = 0
        dc = np.abs(c - g[0])
        print("  run %d: max |dLL| %.3g (pairs %s)   max |dcount| %.3g over %d transitions (%d differ by > 1e-4 rel)" %
              (r, np.nanmax(dll), [int(k) for k in np.nonzero(~(dll <= 2e-5 + 2e-6 * np.abs(g[2])))[0]], dc.max(), len(c),
               int(np.sum(dc > 1e-6 + 1e-4 * np.abs(g[0])))))
    for k, (c, s, ll) in enumerate(out["single"]):
        print("  pair %d alone: LL %.10g (generic in batch %.10g)" % (k, ll[0], g[2][k]))
    same = all(np.array_equal(out[capi.KERNEL_AUTO][0][0], r[0]) and np.array_equal(out[capi.KERNEL_AUTO][0][2], r[2]) for r in out[capi.KERNEL_AUTO][1:-1])
    print("  three runs identical:", same)


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5229
    if "child" in sys.argv:
        child(seed)
        sys.exit(0)
    for name, env in VARIANTS:
        e = dict(os.environ); e.update(env); e["MB_SMALL_JIT_VERBOSE"] = "1"; e["MB_JIT_CACHE"] = "0"
        print("== %s   %s" % (name, {k: (v if len(v) < 40 else v[:37] + "...") for k, v in env.items()}), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(seed), "child"], env=e, capture_output=True, text=True, timeout=900)
        print(r.stdout, end="")
        for line in r.stderr.splitlines():
            if "mode 3" in line or "rror" in line:
                print("  " + line.strip())
        sys.stdout.flush()

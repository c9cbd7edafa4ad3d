"""Build the HIP shared library in-tree (machineboss_amd/libmbhip.so) for gfx950.

``python -m machineboss_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmbhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-fast-math",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    c = os.environ.get("HIPCC")
    if c:
        return c
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def sources():
    return [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "mbhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in sources():
        o = os.path.join(bdir, s + ".o")
        cmd = [_hipcc(), "-x", "hip", "-c", os.path.join(CSRC, s), "-o", o, "-I", os.path.join(ROOT, "include"), "-I", CSRC] + FLAGS + os.environ.get("MB_BUILD_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    subprocess.check_call([_hipcc(), "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950", "-lhiprtc"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Machine + sequence pairs in the text form tests/cxx/test_glue.cpp and tests/cxx/dropin.cpp read (the mock of the reference's
types carries plain numeric weights): used by tests/test_cxx_glue.py, tests/test_dropin.py and bench.py's extra.dropin."""
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def file_weights(em):
    """EvaluatedMachine::init takes log(weight) of the weight it is given: the file carries w = exp(logWeight) at 17 digits and
    the returned machine keeps libm's log(w) (math.log, the function std::log calls), so both sides hold identical doubles."""
    ws = [float("%.17g" % math.exp(l)) for l in em.logWeight]
    em2 = em.withLogWeights(np.array([math.log(w) if w > 0 else -math.inf for w in ws]))
    em2._fileWeights = ws
    return em2


def write_case(path, em, names, pairs, seed=0):
    sym_in = em.inputTokenizer.tok2sym
    sym_out = em.outputTokenizer.tok2sym
    with open(path, "w") as f:
        f.write("%d\n" % em.nStates)
        for s in range(em.nStates):
            a, b = int(em.transOffset[s]), int(em.transOffset[s + 1])
            f.write("%s %d\n" % (names[s], b - a))
            for e in range(a, b):
                f.write("%d %s %s %.17g\n" % (em.dst[e], sym_in[em.inTok[e]] if em.inTok[e] else "-", sym_out[em.outTok[e]] if em.outTok[e] else "-", em._fileWeights[e]))
        f.write("%d\n" % len(pairs))
        for k, (x, y) in enumerate(pairs):
            f.write("in%d out%d %d %s\n%d %s\n" % (k, k, len(x), " ".join(sym_in[t] for t in x), len(y), " ".join(sym_out[t] for t in y)))
        f.write("%d\n" % seed)


def build_exe(outdir, name, opt="-O1"):
    """g++ of tests/cxx/<name>.cpp against the shim and the mock of the reference's types, linked with libmbhip.so."""
    exe = os.path.join(str(outdir), name)
    libdir = os.path.join(ROOT, "machineboss_amd")
    subprocess.check_call(["g++", "-std=c++14", opt, "-Wall", "-Werror", "-DMB_GLUE_MOCK", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "cxx"),
                           "-I", os.path.join(ROOT, "tests", "cxx"), os.path.join(ROOT, "tests", "cxx", name + ".cpp"), "-o", exe,
                           "-L", libdir, "-lmbhip", "-Wl,-rpath," + libdir])
    return exe

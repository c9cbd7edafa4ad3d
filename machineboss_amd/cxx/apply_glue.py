"""Apply the HIP glue to a Machine Boss source tree.

    python machineboss_amd/cxx/apply_glue.py <machineboss tree> --overlay <dir>     # a patched VIEW of the tree
    python machineboss_amd/cxx/apply_glue.py <machineboss tree> --in-place          # what a maintainer commits

What it does to ``src/``:

* ``dpmatrix.h  forward.h  viterbi.h  backward.h`` -- the DP classes of SURVEY.md section 8(a) -- become one-line headers
  that include ``hipdp.h`` (same include guards, same transitive includes), and their ``*.defs.h`` / ``*.cpp`` files
  (``dpmatrix.defs.h forward.defs.h forward.cpp viterbi.cpp backward.cpp``) drop out of the build;
* ``counts.h`` keeps everything except ``struct MachineCounts`` (src/counts.h:11-25), which now comes from ``hipdp.h``:
  ``MachineObjective`` (the M-step, src/counts.h:28-39) stays declared exactly as it was, and in ``counts.cpp`` only the
  members up to ``writeParamCountsJson``/``paramCounts`` (src/counts.cpp:23-106) go -- the shim's MachineCounts carries them;
* ``hipdp.h``, ``mb_dp.hpp``, ``mbhip.h`` are added.

``--prefetch`` additionally adds ONE line in front of each of the two per-pair loops of ``target/boss.cpp`` (``--loglike``,
:796; ``--viterbi / --align``, :826): ``MachineBossHIP::prefetch (eval, data.seqPairs, ...)`` runs the whole SeqPairList as one
batched device call, and the matrices the unchanged loops then construct pair by pair pick their results up (mb_dp.hpp).
Without it every pair is a device call of its own (correct, and latency-bound).

``--overlay`` builds the result as a directory of symlinks to the original files plus the generated headers, so a read-only
tree can be checked: the compiler resolves ``#include "x.h"`` next to the including file's path as named, i.e. inside the
overlay.  Nothing of the original tree is copied except the text of counts.h minus one struct, and only into the overlay.
"""
from __future__ import annotations

import argparse
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

REPLACED = {   # header -> (include guard, includes kept for callers that relied on them transitively)
    "dpmatrix.h": ("DPMATRIX_INCLUDED", ["eval.h", "seqpair.h", "logsumexp.h", "logger.h"]),
    "forward.h": ("FORWARD_INCLUDED", ["dpmatrix.h"]),
    "viterbi.h": ("VITERBI_INCLUDED", ["dpmatrix.h"]),
    "backward.h": ("BACKWARD_INCLUDED", ["forward.h", "counts.h"]),
}
DROPPED = ["dpmatrix.defs.h", "forward.defs.h", "forward.cpp", "viterbi.cpp", "backward.cpp"]
# includes that need GSL / Boost: left out when the result is only parsed (-DHIPDP_SYNTAX_ONLY), e.g. on a box without them
HEAVY = {"logsumexp.h", "logger.h"}


def replaced_header(name: str) -> str:
    guard, incs = REPLACED[name]
    lines = ["#ifndef %s" % guard, "#define %s" % guard, "// replaced by the HIP engine's glue (apply_glue.py): the classes this header declared come from hipdp.h"]
    light = [i for i in incs if i not in HEAVY]
    heavy = [i for i in incs if i in HEAVY]
    lines += ['#include "%s"' % i for i in light]
    if heavy:
        lines += ["#ifndef HIPDP_SYNTAX_ONLY"] + ['#include "%s"' % i for i in heavy] + ["#endif"]
    lines += ['#include "hipdp.h"', "#endif /* %s */" % guard, ""]
    return "\n".join(lines)


def patched_counts_h(text: str) -> str:
    """counts.h without struct MachineCounts; hipdp.h is included before the namespace opens."""
    m = re.search(r"struct\s+MachineCounts\s*\{.*?\n\};\n", text, re.S)
    if not m:
        raise SystemExit("counts.h: struct MachineCounts not found")
    body = text[:m.start()] + "// struct MachineCounts: from hipdp.h (HIP engine)\n" + text[m.end():]
    last = list(re.finditer(r'^#include\s+"[^"]+"\s*$', body, re.M))[-1]
    return body[:last.end()] + '\n#include "hipdp.h"' + body[last.end():]


def patched_counts_cpp(text: str) -> str:
    """counts.cpp without the MachineCounts members (constructors ... paramCounts): the M-step stays."""
    a = text.index("MachineCounts::MachineCounts()")
    b = text.index("WeightExpr makeSquareFunc")
    return text[:a] + "// MachineCounts members: in hipdp.h / mb_dp.hpp (HIP engine)\n\n" + text[b:]


PREFETCH_SITES = (   # (the `if` that opens the block, what the loop reads)
    ('if (vm.count("loglike")) {', "MachineBossHIP::PrefetchLogLike"),
    ('if (vm.count("align") || vm.count("viterbi")) {', "MachineBossHIP::PrefetchViterbi"),
)
EVAL_LINE = "const EvaluatedMachine eval (machine, params);"


def patched_boss_cpp(text: str) -> str:
    """target/boss.cpp with the prefetch line behind the `const EvaluatedMachine eval (machine, params);` of the --loglike block
    (target/boss.cpp:793-794) and of the --viterbi / --align block (:819-822).  Nothing else changes."""
    for opener, what in PREFETCH_SITES:
        a = text.find(opener)
        if a < 0:
            raise SystemExit("boss.cpp: block '%s' not found" % opener)
        b = text.find(EVAL_LINE, a)
        if b < 0 or b - a > 400:
            raise SystemExit("boss.cpp: '%s' not found behind '%s'" % (EVAL_LINE, opener))
        eol = text.index("\n", b)
        indent = text[text.rfind("\n", 0, b) + 1:b]
        text = text[:eol + 1] + indent + "MachineBossHIP::prefetch (eval, data.seqPairs, %s);   // HIP engine: one batched device call for the loop below\n" % what + text[eol + 1:]
    return text


def generated(src: str) -> dict:
    out = {n: replaced_header(n) for n in REPLACED}
    out["counts.h"] = patched_counts_h(open(os.path.join(src, "counts.h")).read())
    out["counts.cpp"] = patched_counts_cpp(open(os.path.join(src, "counts.cpp")).read())
    for name, path in (("hipdp.h", os.path.join(HERE, "hipdp.h")), ("mb_dp.hpp", os.path.join(HERE, "mb_dp.hpp")),
                       ("mb_logsumexp.hpp", os.path.join(HERE, "mb_logsumexp.hpp")), ("mbhip.h", os.path.join(ROOT, "include", "mbhip.h"))):
        out[name] = open(path).read()
    return out


def overlay(tree: str, dest: str, prefetch: bool = False) -> str:
    """A view of `tree` with the glue applied: symlinks + generated files.  Returns dest."""
    tree = os.path.abspath(tree)
    if os.path.exists(dest):
        shutil.rmtree(dest)
    gen = generated(os.path.join(tree, "src"))
    for sub in ("src", os.path.join("t", "src"), "target"):
        d = os.path.join(tree, sub)
        if not os.path.isdir(d):
            continue
        os.makedirs(os.path.join(dest, sub), exist_ok=True)
        for f in os.listdir(d):
            if sub == "src" and (f in gen or f in DROPPED):
                continue
            os.symlink(os.path.join(d, f), os.path.join(dest, sub, f))
    for f, text in gen.items():
        with open(os.path.join(dest, "src", f), "w") as fh:
            fh.write(text)
    if os.path.isdir(os.path.join(tree, "ext")):
        os.symlink(os.path.join(tree, "ext"), os.path.join(dest, "ext"))
    boss = os.path.join(tree, "target", "boss.cpp")
    if prefetch and os.path.exists(boss):
        os.remove(os.path.join(dest, "target", "boss.cpp"))
        with open(os.path.join(dest, "target", "boss.cpp"), "w") as fh:
            fh.write(patched_boss_cpp(open(boss).read()))
    return dest


def in_place(tree: str, prefetch: bool = False) -> None:
    src = os.path.join(tree, "src")
    boss = os.path.join(tree, "target", "boss.cpp")
    if prefetch and os.path.exists(boss):
        text = patched_boss_cpp(open(boss).read())
        with open(boss, "w") as fh:
            fh.write(text)
    for f, text in generated(src).items():
        with open(os.path.join(src, f), "w") as fh:
            fh.write(text)
    for f in DROPPED:
        if os.path.exists(os.path.join(src, f)):
            os.remove(os.path.join(src, f))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("tree")
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--overlay", metavar="DIR")
    g.add_argument("--in-place", action="store_true")
    ap.add_argument("--prefetch", action="store_true", help="also add the one-line batch prefetch in front of the --loglike and --viterbi/--align loops of target/boss.cpp")
    a = ap.parse_args(argv)
    if a.in_place:
        in_place(a.tree, a.prefetch)
    else:
        print(overlay(a.tree, a.overlay, a.prefetch))
    return 0


if __name__ == "__main__":
    sys.exit(main())

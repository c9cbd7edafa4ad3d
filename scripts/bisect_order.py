"""Find the test that makes a later test fail when both run in one process (order-dependent state).
usage: python scripts/bisect_order.py <victim substring> [skip substrings...]"""
import subprocess, sys
victim_key = sys.argv[1]; skip = sys.argv[2:]
out = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu"], capture_output=True, text=True).stdout
ids = [l.strip() for l in out.splitlines() if "::" in l]
vi = next(i for i, t in enumerate(ids) if victim_key in t)
victim = ids[vi]
cands = [t for t in ids[:vi] if not any(s in t for s in skip)]
print("victim", victim, "candidates", len(cands), flush=True)


def fails(subset):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + subset + [victim], capture_output=True, text=True)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    bad = ("FAILED " + victim.split("::")[0]) in r.stdout and victim.split("::")[-1] in "".join(l for l in r.stdout.splitlines() if l.startswith("FAILED"))
    print("  ran %d + victim: %s -> victim %s" % (len(subset), tail, "FAILS" if bad else "passes"), flush=True)
    return bad


if not fails(cands):
    print("victim passes behind all candidates: the culprit is among the skipped tests"); sys.exit(0)
lo, hi = 0, len(cands)
while hi - lo > 1:
    mid = (lo + hi) // 2
    if fails(cands[lo:mid]): hi = mid
    else: lo = mid
print("culprit:", cands[lo])
fails([cands[lo]])

"""The figures the documents quote are GENERATED from the committed evidence ("records drifting from their summaries"):
DESIGN.md section 4.5 from profiles/r06_bench.json (scripts/design_numbers.py), the round's section of profiles/README.md from the
CSV / JSON files beside it (scripts/profiles_readme.py).  These tests regenerate both and compare with what is committed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), "r06"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip()


def test_design_numbers_are_the_committed_bench_line():
    out = _run("design_numbers.py")
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, z = "<!-- NUMBERS:BEGIN -->", "<!-- NUMBERS:END -->"
    assert a in txt and z in txt
    assert txt[txt.index(a) + len(a):txt.index(z)].strip() == out
    assert len(txt.encode()) <= 40 * 1024      # (the current design stays one readable file; history lives under docs/history)


def test_profiles_readme_section_is_generated_from_the_files():
    out = _run("profiles_readme.py")
    txt = open(os.path.join(ROOT, "profiles", "README.md")).read()
    a, z = "<!-- r06:BEGIN", "<!-- r06:END -->"
    assert a in txt and z in txt
    assert txt[txt.index(a):txt.index(z) + len(z)].strip() == out
    for needle in ("r06_bench.json", "r06_bench_kernel_stats.csv", "r06_onetape_sq.json", "r06_kernel_sha.json"):
        assert needle in out and os.path.exists(os.path.join(ROOT, "profiles", needle))

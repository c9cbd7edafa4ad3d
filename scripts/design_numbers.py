"""The "current numbers" table of DESIGN.md section 4.5, generated from the last full bench line (profiles/<tag>_bench.json), so that
the document quotes ONE figure per mode and says where it comes from.  usage: python scripts/design_numbers.py [tag=r06] [--write]
--write replaces the text between <!-- NUMBERS:BEGIN --> and <!-- NUMBERS:END --> in DESIGN.md."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r06")
b = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
e = b["extra"]
rows = []


def iss(r):
    i = (r or {}).get("issue")
    return "%.2f" % i["valu_issue_frac"] if i else "--"


def row(cfg, mode, fam, val, unit, frac, bound, issue, note=""):
    rows.append("| %s | %s | %s | **%s** %s | %s | %s | %s | %s |" % (cfg, mode, fam, val, unit, ("%.3f" % frac) if frac is not None else "--", bound, issue, note))


rf = b["roofline"]
row("4a (headline)", "materialised Forward, 256 x 487 x 10 kb", "tiled", b["value"], "G cells/s", rf["frac"], "HBM, 8 B/cell", iss(rf), "%.0f us per launch, %d launches per step; traffic %s" % (rf["avg_launch_us"], rf["launches_per_step"], ("%.2f B/cell (recorded, kernel hash verified)" % (rf["traffic"] / max(rf["algorithmic_bytes_per_launch"], 1) * 8)) if rf.get("traffic") else "not quoted (see recorded_constants.refused)"))
row("4a", "rolling Forward (`--loglike`), same batch", "tiled", e.get("rolling_gcells_per_gpu"), "G cells/s", None, "latency at 2 wavefronts per SIMD", iss(e.get("rolling_rate")), "no matrix")
v4 = e["viterbi4"]; row("4a", "Viterbi fill / with the paths on the host, 256 pairs", "tiled", "%s / %s" % (v4["fill_only"], v4["value"]), "G cells/s", v4["roofline"]["frac"], "1 B/cell; latency", iss(v4["roofline"]), "one traceback byte per cell")
c4 = e["counts4"]; row("4a", "Backward fill + count sweep (`--train` E-step), 63 pairs", "tiled", c4["value"], "G lattice-cells/s", c4["roofline"]["frac"], "HBM, 16 B/lattice cell", iss(c4["roofline"]), "%.0f ms device" % c4["device_ms"])
cb = e.get("config4b", {})
if "forward_materialised" in cb:
    row("4b (literal composition, 482 states)", "materialised Forward 256 pairs / rolling", "tiled", "%s / %s" % (cb["forward_materialised"], cb["forward_rolling"]), "G cells/s", cb["roofline"]["frac"], "HBM, 8 B/cell", iss(cb["roofline"]), "composed on the box")
    row("4b", "Viterbi fill / with paths (64 pairs); counts (24 pairs)", "tiled", "%s / %s; %s" % (cb["viterbi_fill"], cb["viterbi_with_paths"], cb["counts_lattice"]), "G (lattice-)cells/s", cb["roofline_counts"]["frac"], "counts: 16 B/lattice cell", iss(cb["roofline_counts"]), "counts: 7 columns = 7 wavefronts per CU (history 4.1d)")
c3 = e["counts"]; row("3", "Backward fill + count sweep, 1024 x 400 x 400 (per GPU)", "small", c3["value"], "G lattice-cells/s", c3["roofline"]["frac"], "HBM, 16 B", iss(c3["roofline"]), "%.2f ms device" % c3["device_ms"])
f3 = e["forward_config3"]; row("3", "materialised Forward", "small", f3["value"], "G cells/s", f3["roofline"]["frac"], "HBM, 8 B", iss(f3["roofline"]))
v2 = e["viterbi"]; row("2", "Viterbi fill / with paths, 1024 x 1 kb x 1 kb", "small", "%s / %s" % (v2["fill_only"], v2["value"]), "G cells/s", v2["roofline"]["frac"], "1 B/cell; issue", "--")
f2 = e["forward_config2"]; row("2", "Forward rolling / materialised", "small", "%s / %s" % (f2["rolling"], f2["materialised"]), "G cells/s", f2["roofline"]["frac"], "HBM, 8 B (materialised)", "--")
c1 = e["config1_latency"]; row("1", "one 50 x 50 pair, `--loglike`", "small", c1["warm_call_us"], "us per call", None, "launch latency", "--", "cold start %.0f ms" % c1["cold_start_ms"])
c5 = e.get("config5", {})
if "full_size" in c5:
    fs = c5["full_size"]
    if "viterbi_with_paths" in fs:
        row("5 (64 x 50 kb, 5 063 states)", "rolling Forward / Viterbi fill / with paths (traceback codes)", "one-tape", "%s / %s / %s" % (fs["forward_rolling"], fs["viterbi_fill"], fs["viterbi_with_paths"]), "G cells/s", None,
            "latency chain of 5 rounds per column; k workgroups per sequence, generated code", "see 4.4", "%.0f / %.0f / %.0f ms" % (fs["forward_ms"], fs["viterbi_ms"], fs["viterbi_with_paths_ms"]))
    else:
        row("5 (64 x 50 kb, 5 063 states)", "rolling Forward / Viterbi fill", "one-tape", "%s / %s" % (fs["forward_rolling"], fs["viterbi_fill"]), "G cells/s", None, "latency chain of 5 rounds per column; k workgroups per sequence, generated code", "see 4.4", "%.0f / %.0f ms" % (fs["forward_ms"], fs["viterbi_ms"]))
    if "counts_lattice" in fs:
        row("5 (64 x 50 kb)", "E-step (fills with the fp64 correction term + count kernel)", "one-tape", fs["counts_lattice"], "G lattice-cells/s", None, "the two fills side by side, 2 workgroups per sequence each, then the count kernel (4.4)", "--", "%.0f ms" % fs["counts_ms"])
    row("5 (64 x 2 kb)", "Forward / Viterbi fill / with paths / E-step", "one-tape", "%s / %s / %s / %s" % (c5["forward_rolling"], c5["viterbi_fill"], c5["viterbi_with_paths"], c5["counts_lattice"]), "G (lattice-)cells/s", None, "one workgroup per sequence: vector issue", "--", "256 x 4 kb: %s / %s" % (c5["all_cus"]["forward_rolling"], c5["all_cus"]["viterbi_fill"]))
tr = e.get("train", {})
if "config3" in tr:
    t3 = tr["config3"]["steady"]
    row("3 (`boss --train`, N = 1)", "one Baum-Welch iteration end to end", "small", t3["iteration_ms"], "ms", None, "host algebra + E-step", "--",
        "eval %.1f + set_weights %.1f + E-step %.1f (device %.1f) + M-step %.1f ms; %s G lattice-cells/s end to end" % (t3["eval_ms"], t3["set_weights_ms"], t3["estep_ms"], t3["estep_device_ms"], t3["mstep_ms"], tr["config3"]["lattice_gcells_per_s_end_to_end"]))
if "config5_2kb" in tr:
    t5 = tr["config5_2kb"]["steady"]
    row("5 (64 x 2 kb, `--train`, N = 1)", "one Baum-Welch iteration end to end (M-step: BFGS on the host)", "one-tape", t5.get("iteration_ms", t5["device_facing_ms"]), "ms", None, "host BFGS + E-step", "--",
        "eval %.1f + set_weights %.1f + E-step %.1f (device %.1f) + M-step %.0f ms" % (t5["eval_ms"], t5["set_weights_ms"], t5["estep_ms"], t5["estep_device_ms"], t5.get("mstep_ms", float("nan"))))
dr = e.get("dropin", {})
for key, name in (("config4", "4a through the reference's call sites (8 pairs)"), ("config2", "2 through the reference's call sites (1024 pairs)")):
    d = dr.get(key) or {}
    if "unchanged_loop" in d:
        u, pf, bc = d["unchanged_loop"], d["unchanged_loop_prefetch"], d["batch_c_abi"]
        row(name, "`--loglike`, `--align` loops of boss.cpp: unchanged / + prefetch line / batch C-ABI", "--",
            "%s, %s / %s, %s / %s, %s" % (u["loglike_pairs_per_s"], u["align_pairs_per_s"], pf["loglike_pairs_per_s"], pf["align_pairs_per_s"], bc["loglike_pairs_per_s"], bc["align_pairs_per_s"]),
            "pairs/s", None, "host buffers in, host objects out", "--", "no fp64 matrix over PCIe (fetches: %d)" % (u["matrix_fills"] + pf["matrix_fills"]))
cpu = b.get("cpu_baseline") or {}
txt = ["Source: `profiles/%s_bench.json` (one `python3 bench.py` on the MI355X box; `frac` = algorithmic bytes / device time / 8 TB/s; `issue` = vector-issue fraction from the kernels' ISA, `profiles/%s_valu_model.json`; recorded constants only while the kernel hashes match `profiles/%s_kernel_sha.json`).  Boxes differ by +-5 %%." % (tag, tag, tag), "",
       "| Config | Mode | Family | Rate | HBM frac | Bound | Issue | Note |", "|---|---|---|---|---|---|---|---|"] + rows
if cpu:
    txt += ["", "CPU baseline of the same run: %s %s on %d host cores (%s)." % (cpu["value"], cpu["unit"], cpu["cores"], cpu["kind"])]
out = "\n".join(txt)
print(out)
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
    a, z = s.index("<!-- NUMBERS:BEGIN -->") + len("<!-- NUMBERS:BEGIN -->"), s.index("<!-- NUMBERS:END -->")
    open(p, "w").write(s[:a] + "\n" + out + "\n" + s[z:])

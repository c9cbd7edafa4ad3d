"""Vector-issue model of the run-time specialised kernels ("bound the issue-bound modes with a number").

usage: python scripts/valu_model.py <dir with sources written by MB_MEDIUM_JIT_DUMP / MB_SMALL_JIT_DUMP> [out.json]

Every source is cross-compiled to gfx950 ISA (no GPU needed) and the instructions of its STEP LOOP -- the largest loop of the
kernel -- are counted by class.  Issue weights (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost' and the SIMD-32 note:
a wave64 VALU instruction takes 2 cycles of its SIMD, fp64 arithmetic and transcendentals twice that): plain 32-bit VALU 1,
fp64 VALU 2, v_exp/v_log/v_rcp 2.  Per wave-step a kernel finalises JG supercells of JS cells, so
    issue_slots_per_cell = sum(weight x count) / (JG x JS)
and, with a measured rate R cells/s,
    valu_issue_frac = R x issue_slots_per_cell / (256 CUs x 4 SIMDs x 2.4e9 Hz / 2 cycles per slot).
bench.py reads the JSON (a recorded constant, like the PMC traffic) and multiplies by the rates it measures."""
import collections, glob, json, os, re, subprocess, sys

PEAK_SLOTS = 256 * 4 * 2.4e9 / 2


def loop_stats(asm_path, kernel):
    lines = open(asm_path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(kernel + ":")][0]
    end = [i for i, l in enumerate(lines) if l.strip().startswith("s_endpgm") and i > start][0]
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    a, b = max(loops, key=lambda x: x[1] - x[0])
    c = collections.Counter(); slots = 0.0
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l[0] in ";." or l.split()[0].endswith(":"): continue
        op = re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", l.split()[0])
        if op.startswith("v_"):
            c["valu"] += 1
            w = 2.0 if (("f64" in op and not op.startswith("v_cvt_f32_f64")) or op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32")) else 1.0
            if op.startswith("v_cvt") and "f64" in op: w = 2.0
            slots += w
            if op in ("v_exp_f32", "v_log_f32"): c["transcendental"] += 1
            if "f64" in op: c["fp64"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
    return dict(c), slots


def main():
    src_dir = sys.argv[1]; out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(src_dir, "valu_model.json")
    res = {}
    for src in sorted(glob.glob(os.path.join(src_dir, "*.hip"))):
        if src.endswith("_full.hip"): continue
        text = open(src).read()
        defs = dict(re.findall(r"^#define (J\w+) (\S+)", text, re.M))
        small = "JKERNEL" in defs
        kernel = defs["JKERNEL"] if small else "k_medium_jit"
        full = src[:-4] + "_full.hip"; asm = src[:-4] + ".s"
        open(full, "w").write("#include <hip/hip_runtime.h>\n" + text)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", asm, full],
                           stderr=subprocess.PIPE, text=True)
        if r.returncode: print("compile failed:", src, r.stderr[-500:]); continue
        c, slots = loop_stats(asm, kernel)
        if small:
            cells = 64 * int(defs["JS"]) * 2        # a wavefront is a strip of 64 columns; the loop body is two steps
            note = "small family: loop body = 2 steps of 64 supercells"
        else:
            cells = int(defs["JG"]) * int(defs["JS"])
            note = "tiled family: loop body = 1 step of JG supercells per wavefront"
        vg = [l for l in open(asm) if ".vgpr_count" in l]
        res[os.path.basename(src)] = {"kernel": kernel, "JG": defs.get("JG"), "JC": defs.get("JC"), "JS": defs.get("JS"), "JWAVES": defs.get("JWAVES"), "JMODE": defs.get("JMODE"), "JMAT": defs.get("JMAT"),
                                      "vgprs": int(vg[0].split()[-1]) if vg else None, "loop": c, "issue_slots_per_loop": slots, "cells_per_loop": cells,
                                      "issue_slots_per_cell": slots / cells, "cells_per_s_at_full_issue": PEAK_SLOTS * cells / slots, "note": note}
        print("%-60s valu %4d  slots %6.0f  cells/loop %6d  slots/cell %.3f  -> %.0f G cells/s at 100 %% vector issue" % (os.path.basename(src), c.get("valu", 0), slots, cells, slots / cells, PEAK_SLOTS * cells / slots / 1e9))
    json.dump({"peak_issue_slots_per_s": PEAK_SLOTS, "weights": "32-bit VALU 1, fp64 VALU / f64 conversions 2, v_exp/v_log/v_rcp 2 (MI355X_MICROARCH.md)", "kernels": res}, open(out, "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()

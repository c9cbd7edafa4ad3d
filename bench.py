#!/usr/bin/env python3
"""bench.py -- Giga DP-cells/sec (Forward) on the composed protpsw machine, 1/2/4/8 MI355X.

Headline workload (BASELINE.json configs[3], reading fixed in DESIGN.md): preset/psw2dna.json -- the shipped
GeneWise-style composition of protpsw with the codon model, 271 states / 1684 transitions -- `--use-defaults`
parameters, 256 pairs of a 487-residue synthetic protein (the PF00516 profile length) against 10 kb of synthetic DNA per
GPU (seed 1000*4 + k).  One step = one Forward fill over the whole batch, inputs resident in HBM.

  --mode materialise (default): ForwardMatrix semantics -- every cell is written once to HBM as fp64 (8 algorithmic
        bytes per cell); the 2.7 TB of matrices per step are produced in sub-batches that fit
        the 288 GB of one GPU.
  --mode rolling: RollingOutputForwardMatrix semantics (`boss --loglike`), log-likelihood only, ~0 algorithmic bytes.
  --scaling weak (default): --pairs pairs PER GPU;  strong: --pairs pairs in total, dealt to the ranks by cell count.

Prints ONE JSON line (rank 0).  Besides the contract's keys it carries `roofline`, `cpu_baseline` and, at N = 1, `extra`:
the other modes of the path on the BASELINE configs they are quoted on (config 3: Forward-Backward counts on protpsw,
config 2: Viterbi + traceback on dnapsw, config 1: cold-start latency of one 50-aa pair), each with its own roofline
block; the other two modes of the HEADLINE machine on config 4's own shape (`viterbi4`: one traceback byte per cell,
`counts4`: no Forward matrix); config 5 at its stated 64 x 50 kb; and `nonuniform`, the second parameter set SURVEY 8(d) asks
for (a Baum-Welch fit).  At N > 1 `extra.em_iteration` times the one collective of the path: an E-step on the rank's shard
of config 3 followed by the RCCL all-reduce of the counts, and `per_rank` lists every rank's cells and seconds.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts N ranks itself (torch.distributed.run as a child
process; this process has not touched the GPU at that point) and relays rank 0's line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured achievable copy rate
TRANSCENDENTAL_PEAK_T = 157.0 / 2 / 4   # v_exp_f32 / v_log_f32 per second (x 1e12): fp32 vector peak 157 TFLOP/s = 78.5 T lane-ops/s, transcendentals at quarter rate (MI355X_MICROARCH.md)
BYTES_PER_CELL = 8      # materialised Forward: one fp64 store per cell (w = 8)


PROFILE_TAG = "r06"      # profiles/<tag>_*: the recorded constants this bench line may quote
DUMP_DIR = None          # generated kernel sources of THIS run (MB_MEDIUM_JIT_DUMP / MB_SMALL_JIT_DUMP), hashed against profiles/<tag>_kernel_sha.json
REFUSED = {}             # recorded figure -> why it was not quoted


def recorded(name):
    """A committed measurement under profiles/ (None when absent)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def dump_kernels_as(tag):
    """From here on the library writes the source of every kernel it generates to DUMP_DIR/<tag>.<kind>.hip (read when a kernel is built)."""
    if DUMP_DIR:
        os.environ["MB_MEDIUM_JIT_DUMP"] = os.path.join(DUMP_DIR, tag)
        os.environ["MB_SMALL_JIT_DUMP"] = os.path.join(DUMP_DIR, tag)
        os.environ["MB_WIDE_JIT_DUMP"] = os.path.join(DUMP_DIR, tag)


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def kernel_sha_now():
    """sha256[:16] of every generated kernel source of this run, and of the ahead-of-time one-tape kernels' source files."""
    out = {}
    if DUMP_DIR and os.path.isdir(DUMP_DIR):
        for f in sorted(os.listdir(DUMP_DIR)):
            if f.endswith(".hip"):
                out[f[:-4]] = _sha16(os.path.join(DUMP_DIR, f))
    import hashlib
    h = hashlib.sha256()
    for f in ("mb_wide.hip", "mb_wide.h", "mb_wide_jit.cpp", "mb_wide_jit_src.h"):      # the one-tape family: interpreter kernels + the generator of the per-machine sweep
        h.update(open(os.path.join(ROOT, "machineboss_amd", "csrc", f), "rb").read())
    out["aot:mb_wide"] = h.hexdigest()[:16]
    return out


def kernels_unchanged(what, kernels):
    """True when every kernel a recorded figure was measured on still has the source it had when profiles/<tag>_kernel_sha.json was
    written (by `python bench.py --write-kernel-sha`, in the round's profile run): a recorded PMC traffic / issue figure is quoted
    only then -- otherwise the line carries null and REFUSED says why (constants must not go stale silently)."""
    ref = recorded(PROFILE_TAG + "_kernel_sha.json")
    if not ref:
        REFUSED[what] = "no profiles/%s_kernel_sha.json" % PROFILE_TAG
        return False
    now = kernel_sha_now()
    for k in kernels:
        if k not in now:
            REFUSED[what] = "kernel %s was not generated in this run" % k
            return False
        if ref.get("kernels", {}).get(k) != now[k]:
            REFUSED[what] = "kernel %s changed since profiles/%s_* were recorded (%s -> %s)" % (k, PROFILE_TAG, ref.get("kernels", {}).get(k), now[k])
            return False
    return True


def valu_issue(keys, cells_per_s):
    """Vector-issue fraction of a mode: issue slots per cell of its kernels' step loops, from their gfx950 ISA
    (scripts/valu_model.py: plain VALU 1, fp64 / transcendental 2, one slot = 2 cycles of a SIMD-32) x the measured rate /
    (256 CUs x 4 SIMDs x 2.4 GHz / 2).  None when the model has no entry for a kernel or a kernel changed since it was recorded."""
    model = recorded(PROFILE_TAG + "_valu_model.json")
    if not model:
        REFUSED["issue:" + "+".join(keys)] = "no profiles/%s_valu_model.json" % PROFILE_TAG
        return None
    if not kernels_unchanged("issue:" + "+".join(keys), [k[:-4] if k.endswith(".hip") else k for k in keys]):
        return None
    try:
        per_cell = sum(model["kernels"][k]["issue_slots_per_cell"] for k in keys)
        return {"valu_issue_frac": round(cells_per_s * per_cell / model["peak_issue_slots_per_s"], 4), "issue_slots_per_cell": round(per_cell, 4),
                "at_full_issue_gcells": round(model["peak_issue_slots_per_s"] / per_cell / 1e9, 1), "kernels": list(keys),
                "source": "profiles/%s_valu_model.json (ISA of the generated kernels, scripts/valu_model.py; recorded -- kernel sources verified unchanged by hash)" % PROFILE_TAG}
    except KeyError:
        return None


def pmc_traffic(name, units, kernels):
    """HBM bytes per call of a mode from its committed PMC passes (profiles/<tag>_<name>), scaled to this run's units -- or (None, why)."""
    full = PROFILE_TAG + "_" + name
    p = recorded(full)
    if not p:
        REFUSED["traffic:" + name] = "no profiles/" + full
        return None, None
    if not kernels_unchanged("traffic:" + name, kernels):
        return None, REFUSED["traffic:" + name]
    return round(p["hbm_bytes_per_cell"] * units), "profiles/%s (%.2f B per cell, separate --pmc WRITE_SIZE / FETCH_SIZE passes; recorded -- kernel sources verified unchanged by hash)" % (full, p["hbm_bytes_per_cell"])


def onetape_issue():
    """Issue fraction of the one-tape sweeps at 64 x 50 kb from the round's SQ passes (profiles/<tag>_onetape_sq.json, written by
    scripts/profile_onetape.sh + summarize): SQ_INSTS_VALU of the dispatches / (SIMDs of the occupied CUs x cycles at 2.4 GHz) x 2 cycles
    per wave64 instruction -- quoted only while mb_wide.hip is the file it was measured on (no pasted literals)."""
    p = recorded(PROFILE_TAG + "_onetape_sq.json")
    if not p:
        REFUSED["issue:onetape"] = "no profiles/%s_onetape_sq.json" % PROFILE_TAG
        return None
    if not kernels_unchanged("issue:onetape", ["aot:mb_wide"]):
        return None
    out = {"what": p.get("what"), "source": "profiles/%s_onetape_sq.json (recorded -- mb_wide.hip verified unchanged by hash)" % PROFILE_TAG}
    for k, v in p.get("sweeps", {}).items():
        out[k] = {"valu_issue_frac_unweighted": round(v["SQ_INSTS_VALU"] * 2 / (v["cus"] * 4 * v["seconds"] * 2.4e9), 3), "SQ_WAIT_ANY": v.get("SQ_WAIT_ANY")}
    return out


def host_cores() -> int:
    """CPU threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can see
    256 CPUs and be throttled to 8) and by one socket's worth (the north-star compares with a single socket)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 128))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["materialise", "rolling"], default="materialise")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--pairs", type=int, default=256, help="pairs per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--inlen", type=int, default=487)
    ap.add_argument("--outlen", type=int, default=10000)
    ap.add_argument("--preset", default="psw2dna")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra (non-headline) measurements")
    ap.add_argument("--extra-em-only", action="store_true", help="of the extras, only the EM iteration with its all-reduce (needs a rank group: N > 1 or MB_BENCH_FORCE_COMM=1)")
    ap.add_argument("--write-kernel-sha", metavar="FILE", help="write the hashes of this run's generated kernel sources (profiles/<tag>_kernel_sha.json: what recorded PMC / issue figures are checked against)")
    ap.add_argument("--dropin-only", action="store_true", help="only extra.dropin (the reference's call sites through the C++ classes); prints that block")
    ap.add_argument("--quick", action="store_true", help="with --dropin-only: small batches")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """--gpus N typed directly: N fresh ranks under torch.distributed.run, started BEFORE this process touches the GPU
    (a process that has initialised HIP must never be replaced or re-exec'd on this pool)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


TIMINGS = {}      # label -> [wall ms, device ms] of every timed mode of the line (extra.timings; host_overhead_flags reads it)


def timed(fn, reps=1, label=None):
    """Mean wall time of `reps` calls after one warm-up call.  The results stay referenced until the clock stops: freeing
    a large host array (munmap) inside the timed region stalls the NEXT GPU submission of the process by tens of
    milliseconds on this driver (scripts/vit_timing.py keep / drop), which is the caller's cost, not the library's.
    label: wall AND device milliseconds (HIP events of the library around its kernels, mean over the calls) go to TIMINGS -- round 5
    shipped a mode whose wall clock was 95 x its device time (a 230 GB pool re-allocated in every call) and only the device-side
    figure was printed."""
    from machineboss_amd import capi
    held = [fn()]
    dev = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        held.append(fn())
        dev += capi.last_device_ms()
    dt = (time.perf_counter() - t0) / reps
    if label:
        TIMINGS[label] = [round(dt * 1e3, 3), round(dev / reps, 3)]
    return held[-1], dt


def host_overhead_flags():
    """modes whose wall clock exceeds 1.2 x their device time (modes of more than 50 ms): memory management, host-side planning or
    copies that the device-side roofline figures do not show"""
    return sorted(k for k, (w, d) in TIMINGS.items() if w > 50.0 and w > 1.2 * d)


def extra_single_gpu(capi, np, hbm_peak):
    """The other modes of the hot path on the BASELINE configs they belong to (one GPU, inputs resident in HBM)."""
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_batch, synth_tokens
    out = {}

    def machine(preset):
        m = Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", preset + ".json"))
        em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
        return em

    # config 1: protpsw, ONE 50-aa pair: cold start = machine upload + program build + kernel JIT (or disk-cache hit) + run
    em1 = machine("protpsw")
    dump_kernels_as("protpsw")
    t0 = time.perf_counter()
    dm1 = capi.DeviceMachine(em1)
    b1 = capi.DeviceBatch(dm1, *synth_batch(1, 1, 50, 50, em1.nInTok, em1.nOutTok))
    ll1 = b1.forward(capi.MB_ROLLING)
    cold = time.perf_counter() - t0
    _, warm = timed(lambda: b1.forward(capi.MB_ROLLING), 20, label="config1.warm_call")
    out["config1_latency"] = {"workload": "protpsw, one 50 x 50 aa pair, boss --loglike (rolling Forward)", "cold_start_ms": round(cold * 1e3, 2),
                              "warm_call_us": round(warm * 1e6, 1), "loglike": float(ll1[0]), "kernel": capi.last_kernel_name(),
                              "jit": capi.jit_stats()}

    # config 3 (per GPU): protpsw, 1024 x 400 x 400, Forward-Backward + counts: 16 algorithmic bytes per lattice cell
    # (Backward written once, read once by the count sweep)
    b3 = capi.DeviceBatch(dm1, *synth_batch(3, 1024, 400, 400, em1.nInTok, em1.nOutTok))
    cells3 = b3.cells()
    b3.counts()
    dev = []; t0 = time.perf_counter()
    for _ in range(5):
        cnt, s3, _ = b3.counts(); dev.append(capi.last_device_ms())
    wall = (time.perf_counter() - t0) / 5
    TIMINGS["counts_config3"] = [round(wall * 1e3, 3), round(sum(dev) / len(dev), 3)]
    ach = 16.0 * cells3 / (sum(dev) / len(dev) / 1e3) / 1e9
    tr3, tr3src = pmc_traffic("counts_pmc_hbm.json", cells3, ["protpsw.m0.mat.bwd", "protpsw.m3.roll.fwd"])
    nsym = float(cnt[np.asarray(em1.inTok) != 0].sum()), float(cnt[np.asarray(em1.outTok) != 0].sum())
    out["counts"] = {"workload": "config 3 per GPU: protpsw (8 states, 450 transitions), 1024 pairs x 400 x 400 aa, Backward + Forward/count sweep (MachineCounts)",
                     "value": round(cells3 / wall / 1e9, 2), "unit": "G lattice-cells/s (two matrices per lattice cell)", "ms": round(wall * 1e3, 3),
                     "device_ms": round(sum(dev) / len(dev), 3),
                     "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": hbm_peak, "unit": "GB/s", "frac": round(ach / hbm_peak, 4),
                                  "algorithmic_bytes_per_lattice_cell": 16, "kernel": "k_small_sum_bwd + " + capi.last_kernel_name(), "traffic": tr3, "traffic_source": tr3src,
                                  "issue": valu_issue(["protpsw.m0.mat.bwd.hip", "protpsw.m3.roll.fwd.hip"], cells3 / (sum(dev) / len(dev) / 1e3))},
                     "symbol_count_invariant": [nsym[0] / (1024 * 400), nsym[1] / (1024 * 400)], "loglike_sum": float(s3)}
    mfw, tf = timed(lambda: b3.forward(capi.MB_MATERIALISE), 5, label="forward_config3.materialised"); devf = capi.last_device_ms()
    out["forward_config3"] = {"workload": "protpsw 1024 x 400 x 400, materialised Forward", "value": round(cells3 / tf / 1e9, 2), "unit": "Gcells/s",
                              "roofline": {"bound": "hbm", "achieved": round(8.0 * cells3 / (devf / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                           "frac": round(8.0 * cells3 / (devf / 1e3) / 1e9 / hbm_peak, 4), "kernel": capi.last_kernel_name(),
                                           "traffic": pmc_traffic("forward3_pmc_hbm.json", cells3, ["protpsw.m0.mat.fwd"])[0],
                                           "issue": valu_issue(["protpsw.m0.mat.fwd.hip"], cells3 / (devf / 1e3))}}
    del b3

    # config 2: dnapsw, 1024 x 1 kb x 1 kb: Viterbi with traceback (1 algorithmic byte per cell: the traceback pointer), Forward
    em2 = machine("dnapsw")
    dump_kernels_as("dnapsw")
    dm2 = capi.DeviceMachine(em2)
    b2 = capi.DeviceBatch(dm2, *synth_batch(2, 1024, 1000, 1000, em2.nInTok, em2.nOutTok))
    cells2 = b2.cells()
    (vll, off, edges), tv = timed(lambda: b2.viterbi(paths=True), 3, label="viterbi.with_paths"); devv = capi.last_device_ms(); kv = capi.last_kernel_name()
    _, tvf = timed(lambda: b2.viterbi(paths=False), 3, label="viterbi.fill"); devvf = capi.last_device_ms()
    out["viterbi"] = {"workload": "config 2: dnapsw (8 states, 34 transitions), 1024 pairs x 1000 x 1000 nt, ViterbiMatrix + traceBack",
                      "value": round(cells2 / tv / 1e9, 2), "unit": "Gcells/s (fill + traceback + paths copied to the host)",
                      "fill_only": round(cells2 / tvf / 1e9, 2), "device_ms": round(devv, 3), "fill_device_ms": round(devvf, 3),
                      "path_edges": int(off[-1]), "loglike_sum": float(vll.sum()),
                      "roofline": {"bound": "valu", "note": "1 traceback byte per cell: HBM traffic is 1/8 of the materialised fill; the sweep is bound by vector instruction issue (fp64 add/max at half rate), see DESIGN.md",
                                   "achieved": round(1.0 * cells2 / (devvf / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                   "frac": round(1.0 * cells2 / (devvf / 1e3) / 1e9 / hbm_peak, 4), "algorithmic_bytes_per_cell": 1, "kernel": kv}}
    _, tr = timed(lambda: b2.forward(capi.MB_ROLLING), 3, label="forward_config2.rolling")
    _, tm = timed(lambda: b2.forward(capi.MB_MATERIALISE), 3, label="forward_config2.materialised"); devm = capi.last_device_ms()
    out["forward_config2"] = {"workload": "dnapsw 1024 x 1000 x 1000, Forward", "rolling": round(cells2 / tr / 1e9, 2), "materialised": round(cells2 / tm / 1e9, 2),
                              "unit": "Gcells/s", "roofline": {"bound": "hbm", "achieved": round(8.0 * cells2 / (devm / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                                               "frac": round(8.0 * cells2 / (devm / 1e3) / 1e9 / hbm_peak, 4), "kernel": capi.last_kernel_name()}}
    del b2

    # config 4's own shape for the other two modes of the HEADLINE machine (psw2dna, 487 aa x 10 kb): `--viterbi/--align` keeps ONE
    # traceback byte per cell (1 B), `--train`'s E-step keeps no Forward matrix (16 B per lattice cell: the
    # Backward matrix written once and read once)
    em4 = machine("psw2dna")
    dump_kernels_as("psw2dna")
    dm4 = capi.DeviceMachine(em4)
    nv = 256
    b4 = capi.DeviceBatch(dm4, *synth_batch(4, nv, 487, 10000, em4.nInTok, em4.nOutTok))
    cells4 = b4.cells()
    (v4, off4, e4), tv4 = timed(lambda: b4.viterbi(paths=True), 1, label="viterbi4.with_paths"); devv4 = capi.last_device_ms(); kv4 = capi.last_kernel_name()
    _, tvf4 = timed(lambda: b4.viterbi(paths=False), 1, label="viterbi4.fill"); devvf4 = capi.last_device_ms()
    out["viterbi4"] = {"workload": "config 4: psw2dna (271 states), %d pairs x 487 aa x 10000 nt, ViterbiMatrix + traceBack, one traceback byte per cell" % nv,
                       "value": round(cells4 / tv4 / 1e9, 2), "unit": "Gcells/s (fill + traceback + paths copied to the host)", "fill_only": round(cells4 / tvf4 / 1e9, 2),
                       "device_ms": round(devv4, 3), "fill_device_ms": round(devvf4, 3), "path_edges": int(off4[-1]), "loglike_sum": float(v4.sum()),
                       "roofline": {"bound": "valu", "note": "1 traceback byte per cell (+ boundary records and halo rows: profiles/r03_viterbi4_pmc_hbm.json); the max sweep is bound by vector instruction issue, not by HBM",
                                    "achieved": round(1.0 * cells4 / (devvf4 / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                    "frac": round(1.0 * cells4 / (devvf4 / 1e3) / 1e9 / hbm_peak, 4), "algorithmic_bytes_per_cell": 1, "kernel": kv4,
                                    "traffic": pmc_traffic("viterbi4_pmc_hbm.json", cells4, ["psw2dna.tb.tiles.fwd.exact"])[0],
                                    "issue": valu_issue(["psw2dna.tb.tiles.fwd.exact.hip"], cells4 / (devvf4 / 1e3))}}
    del b4
    nc4 = 63      # three chunks of Backward matrices (21 pairs of 10.6 GB each fit the 80 % budget of a 288 GB GPU)
    b4c = capi.DeviceBatch(dm4, *synth_batch(4, nc4, 487, 10000, em4.nInTok, em4.nOutTok))
    cells4c = b4c.cells()
    (cnt4, s4, _), tc4 = timed(lambda: b4c.counts(), 1, label="counts4"); devc4 = capi.last_device_ms()
    ach4 = 16.0 * cells4c / (devc4 / 1e3) / 1e9
    out["counts4"] = {"workload": "config 4: psw2dna, %d pairs x 487 aa x 10000 nt, Backward fill + Forward/count sweep without a Forward matrix (MachineCounts)" % nc4,
                      "value": round(cells4c / tc4 / 1e9, 2), "unit": "G lattice-cells/s (two matrices per lattice cell)", "ms": round(tc4 * 1e3, 2), "device_ms": round(devc4, 2),
                      "roofline": {"bound": "hbm", "achieved": round(ach4, 1), "peak": hbm_peak, "unit": "GB/s", "frac": round(ach4 / hbm_peak, 4),
                                   "algorithmic_bytes_per_lattice_cell": 16, "kernel": "k_medium_jit (Backward fill) + k_medium_jit (count sweep: closure Forward rounds + flat usage pass)",
                                   "traffic": pmc_traffic("counts4_pmc_hbm.json", cells4c, ["psw2dna.sum.mat.bwd.clos", "psw2dna.cnt.tiles.fwd.clos"])[0],
                                   "traffic_source": pmc_traffic("counts4_pmc_hbm.json", cells4c, ["psw2dna.sum.mat.bwd.clos", "psw2dna.cnt.tiles.fwd.clos"])[1],
                                   "issue": valu_issue(["psw2dna.sum.mat.bwd.clos.hip", "psw2dna.cnt.tiles.fwd.clos.hip"], cells4c / (devc4 / 1e3))},
                      "symbol_count_invariant": [float(cnt4[np.asarray(em4.inTok) != 0].sum()) / (nc4 * 487), float(cnt4[np.asarray(em4.outTok) != 0].sum()) / (nc4 * 10000)],
                      "loglike_sum": float(s4)}
    del b4c

    # config 4 read LITERALLY ("C4b"): protpsw . translate . dnapsw with dnapsw's constraints cleared, composed
    # here (482 states, 3095 transitions, 22 silent levels) -- the three modes on the config's own shape, each with its roofline
    try:
        from machineboss_amd import algebra as A4
        t0 = time.perf_counter()
        em4b = EvaluatedMachine.fromMachine(A4.config4bMachine(os.path.join(ROOT, "tests", "golden", "preset")), None, useDefaults=True)
        tc4b = time.perf_counter() - t0
        dump_kernels_as("c4b")
        dm4b = capi.DeviceMachine(em4b)
        bb = capi.DeviceBatch(dm4b, *synth_batch(4, 256, 487, 10000, em4b.nInTok, 3))      # DNA over {A,C,G}: no stop codons
        cellsb = bb.cells()
        llb, tfb = timed(lambda: bb.forward(capi.MB_MATERIALISE), 1, label="config4b.forward_materialised"); devfb = capi.last_device_ms(); kfb = capi.last_kernel_name()
        _, trb = timed(lambda: bb.forward(capi.MB_ROLLING), 1, label="config4b.forward_rolling")
        del bb
        bbv = capi.DeviceBatch(dm4b, *synth_batch(4, 64, 487, 10000, em4b.nInTok, 3))
        cellsbv = bbv.cells()
        (vb, offb, eb), tvb = timed(lambda: bbv.viterbi(paths=True), 1, label="config4b.viterbi_with_paths")
        _, tvfb = timed(lambda: bbv.viterbi(paths=False), 1, label="config4b.viterbi_fill"); devvfb = capi.last_device_ms()
        del bbv
        bbc = capi.DeviceBatch(dm4b, *synth_batch(4, 24, 487, 10000, em4b.nInTok, 3))      # two chunks of Backward matrices (18.8 GB each)
        cellsbc = bbc.cells()
        (cntb, sb, _), tcb = timed(lambda: bbc.counts(), 1, label="config4b.counts"); devcb = capi.last_device_ms()
        del bbc
        out["config4b"] = {"workload": "config 4 literally: protpsw . translate . dnapsw (dnapsw's constraints cleared; %d states, %d transitions), composed here in %.2f s; 256 / 64 / 24 pairs x 487 aa x 10000 nt" % (em4b.nStates, em4b.nTransitions, tc4b),
                           "forward_materialised": round(cellsb / tfb / 1e9, 2), "forward_rolling": round(cellsb / trb / 1e9, 2), "unit": "Gcells/s",
                           "viterbi_with_paths": round(cellsbv / tvb / 1e9, 2), "viterbi_fill": round(cellsbv / tvfb / 1e9, 2), "path_edges": int(offb[-1]),
                           "counts_lattice": round(cellsbc / tcb / 1e9, 2), "loglike_checksum": float(np.sum(llb)), "viterbi_checksum": float(np.sum(vb)),
                           "symbol_count_invariant": [float(cntb[np.asarray(em4b.inTok) != 0].sum()) / (24 * 487), float(cntb[np.asarray(em4b.outTok) != 0].sum()) / (24 * 10000)],
                           "roofline": {"bound": "hbm", "achieved": round(8.0 * cellsb / (devfb / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                        "frac": round(8.0 * cellsb / (devfb / 1e3) / 1e9 / hbm_peak, 4), "algorithmic_bytes_per_cell": 8, "kernel": kfb, "what": "materialised Forward, 256 pairs",
                                        "traffic": pmc_traffic("forward4b_pmc_hbm.json", cellsb, ["c4b.sum.mat.fwd.clos"])[0],
                                        "issue": valu_issue(["c4b.sum.mat.fwd.clos.hip"], cellsb / (devfb / 1e3))},
                           "roofline_viterbi": {"bound": "valu", "achieved": round(1.0 * cellsbv / (devvfb / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                                "frac": round(1.0 * cellsbv / (devvfb / 1e3) / 1e9 / hbm_peak, 4), "algorithmic_bytes_per_cell": 1},
                           "roofline_counts": {"bound": "hbm", "achieved": round(16.0 * cellsbc / (devcb / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s",
                                               "frac": round(16.0 * cellsbc / (devcb / 1e3) / 1e9 / hbm_peak, 4), "algorithmic_bytes_per_lattice_cell": 16,
                                               "traffic": pmc_traffic("c4b_counts_pmc_hbm.json", cellsbc, ["c4b.sum.mat.bwd.clos", "c4b.cnt.tiles.fwd.clos"])[0],
                                               "issue": valu_issue(["c4b.sum.mat.bwd.clos.hip", "c4b.cnt.tiles.fwd.clos.hip"], cellsbc / (devcb / 1e3))}}
        del dm4b
    except Exception as e:
        out["config4b"] = {"error": str(e)}

    # config 5: HMMER profile . simple_introns . translate . dnapsw assembled here (first 20 nodes of the fn3 profile: 5063
    # states, the "~5k states" of the config), a one-tape generator; 64 sequences x 2 kb (one workgroup per sequence)
    try:
        from machineboss_amd import algebra as A
        from machineboss_amd.hmmer import HmmerModel
        P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
        t0 = time.perf_counter()
        h = HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(20)
        em5 = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
        tc = time.perf_counter() - t0
        dump_kernels_as("config5")
        dm5 = capi.DeviceMachine(em5)
        b5 = capi.DeviceBatch(dm5, *synth_batch(5, 64, 0, 2000, em5.nInTok, em5.nOutTok))
        cells5 = b5.cells()
        ll5, t5 = timed(lambda: b5.forward(capi.MB_ROLLING), 2, label="config5.forward_rolling"); k5 = capi.last_kernel_name()
        _, t5v = timed(lambda: b5.viterbi(paths=False), 1, label="config5.viterbi_fill")
        (_, _, e5), t5p = timed(lambda: b5.viterbi(), 1, label="config5.viterbi_with_paths")
        (cnt5, s5, _), t5c = timed(lambda: b5.counts(), 1, label="config5.counts"); k5c = capi.last_kernel_name()
        out["config5"] = {"workload": "fn3 profile (20 nodes) . simple_introns . translate . dnapsw: %d states, %d transitions, one tape; 64 sequences x 2000 nt" % (em5.nStates, em5.nTransitions),
                          "compose_s": round(tc, 2), "forward_rolling": round(cells5 / t5 / 1e9, 2), "viterbi_fill": round(cells5 / t5v / 1e9, 2),
                          "viterbi_with_paths": round(cells5 / t5p / 1e9, 2), "path_edges": int(len(e5)),
                          "counts_lattice": round(cells5 / t5c / 1e9, 2), "unit": "Gcells/s", "counts_ms": round(t5c * 1e3, 1),
                          "kernels": [k5, k5c], "loglike_sum": float(np.sum(ll5)),
                          "symbol_count_invariant": float(cnt5[np.asarray(em5.outTok) != 0].sum()) / (64 * 2000),
                          "roofline": {"bound": "valu + barriers", "note": "k workgroups per sequence (DESIGN.md 4.2d: 64 sequences x 4 parts, or cut in two x 2 parts -- every CU holds one workgroup); the retimed sweep (4.2b) turns a column's 366 dependent silent levels into a period of 9-10 barrier-separated rounds; the period is bound by that chain of stages; no HBM or MFMA bound applies",
                                       "issue": onetape_issue()}}
        del b5
        # ... with every CU busy: 256 sequences x 4 kb
        b5w = capi.DeviceBatch(dm5, *synth_batch(5, 256, 0, 4000, em5.nInTok, em5.nOutTok))
        cells5w = b5w.cells()
        _, t5w = timed(lambda: b5w.forward(capi.MB_ROLLING), 1, label="config5.all_cus.forward_rolling"); k5w = capi.last_kernel_name()
        _, t5wv = timed(lambda: b5w.viterbi(paths=False), 1, label="config5.all_cus.viterbi_fill"); k5wv = capi.last_kernel_name()
        out["config5"]["all_cus"] = {"workload": "the same machine, 256 sequences x 4000 nt (one workgroup per CU)", "forward_rolling": round(cells5w / t5w / 1e9, 2),
                                     "viterbi_fill": round(cells5w / t5wv / 1e9, 2), "unit": "Gcells/s", "kernels": [k5w, k5wv]}
        del b5w
        # ... and at the config's STATED length: 64 sequences x 50 kb on this one GPU (16.2 G cells per matrix)
        b5f = capi.DeviceBatch(dm5, *synth_batch(5, 64, 0, 50000, em5.nInTok, em5.nOutTok))
        cells5f = b5f.cells()
        ll5f, t5f = timed(lambda: b5f.forward(capi.MB_ROLLING), 1, label="config5.full_size.forward_rolling"); k5f = capi.last_kernel_name()
        (v5f, _, _), t5vf = timed(lambda: b5f.viterbi(paths=False), 1, label="config5.full_size.viterbi_fill"); k5vf = capi.last_kernel_name()
        # --align at that size: one traceback code per cell (16 GB where the fp64 Viterbi matrices would be 130), the paths on the host
        (v5p, o5p, e5p), t5pf = timed(lambda: b5f.viterbi(), 1, label="config5.full_size.viterbi_with_paths"); k5pf = capi.last_kernel_name()
        out["config5"]["full_size"] = {"viterbi_with_paths": round(cells5f / t5pf / 1e9, 2), "viterbi_with_paths_ms": round(t5pf * 1e3, 1), "path_edges": int(len(e5p)),
                                       "paths_kernel": k5pf, "paths_score_equals_fill": bool(np.array_equal(v5p, v5f)),
                                       "note_paths": "the paths are oracle-checked bit for bit at this length in tests/test_gpu_parity.py (test_baseline_config5_one_sequence_at_50kb_against_the_oracle)"}
        out["config5"]["full_size"].update({"workload": "the same machine, 64 sequences x 50000 nt (BASELINE config 5 as stated, all on one GPU)", "cells": int(cells5f),
                                       "forward_rolling": round(cells5f / t5f / 1e9, 2), "forward_ms": round(t5f * 1e3, 1), "viterbi_fill": round(cells5f / t5vf / 1e9, 2),
                                       "viterbi_ms": round(t5vf * 1e3, 1), "unit": "Gcells/s", "kernels": [k5f, k5vf], "loglike_sum": float(np.sum(ll5f)),
                                       "viterbi_le_forward": bool(np.all(v5f <= ll5f + 1e-6 * np.abs(ll5f)))})
        # ... and the E-step at that size (two fp64 matrices of 130 GB each: the library cuts the batch into chunks that fit; fills with the fp64
        # correction term, sequences being >= 10 000 symbols)
        try:
            (cnt5f, s5cf, _), t5cf = timed(lambda: b5f.counts(), 1, label="config5.full_size.counts")
            out["config5"]["full_size"].update({"counts_lattice": round(cells5f / t5cf / 1e9, 2), "counts_ms": round(t5cf * 1e3, 1), "counts_device_ms": round(capi.last_device_ms(), 1),
                                                "counts_kernel": capi.last_kernel_name(), "counts_symbol_invariant": float(cnt5f[np.asarray(em5.outTok) != 0].sum()) / (64 * 50000),
                                                "counts_note": "Forward + Backward fills side by side (k_wide_jit, 2 workgroups per sequence each, fp64 log-sum-exp correction term: sequences >= 10 000 symbols) + k_onetape_counts, one chunk of 64 sequences (2 x 129.6 GB of matrices), G lattice-cells/s"})
        except Exception as e:
            out["config5"]["full_size"]["counts_error"] = str(e)
        del b5f, e5p
        # ... and what ONE GPU holds when the config is split over eight: 8 sequences x 50 kb.  A sequence is serial along its columns, so the
        # only parallelism left is inside a column: k workgroups per sequence, against one workgroup per sequence
        b5e = capi.DeviceBatch(dm5, *synth_batch(5, 8, 0, 50000, em5.nInTok, em5.nOutTok))
        eight = {"workload": "the same machine, 8 sequences x 50000 nt (one GPU's share of config 5 split over 8 GPUs)"}
        parts_env = os.environ.get("MB_ONETAPE_PARTS")      # (restored below: a run under a knob keeps it for the other extras)
        for label, env in (("k_workgroups_per_sequence", parts_env), ("one_workgroup_per_sequence", "1")):
            if env is None: os.environ.pop("MB_ONETAPE_PARTS", None)
            else: os.environ["MB_ONETAPE_PARTS"] = env
            _, te = timed(lambda: b5e.forward(capi.MB_ROLLING), 1, label="config5.eight_per_gpu.%s.forward_rolling" % label); ke = capi.last_kernel_name()
            _, tve = timed(lambda: b5e.viterbi(paths=False), 1, label="config5.eight_per_gpu.%s.viterbi_fill" % label); kve = capi.last_kernel_name()
            _, tpe = timed(lambda: b5e.viterbi(), 1, label="config5.eight_per_gpu.%s.viterbi_with_paths" % label)
            eight[label] = {"forward_ms": round(te * 1e3, 1), "viterbi_ms": round(tve * 1e3, 1), "viterbi_with_paths_ms": round(tpe * 1e3, 1), "kernels": [ke, kve]}
        if parts_env is None: os.environ.pop("MB_ONETAPE_PARTS", None)
        else: os.environ["MB_ONETAPE_PARTS"] = parts_env
        fs = out["config5"]["full_size"]
        eight["strong_scaling_ceiling_over_8_gpus"] = {k: round(fs[k] / eight["k_workgroups_per_sequence"][k], 2) for k in ("forward_ms", "viterbi_ms", "viterbi_with_paths_ms")}
        eight["note"] = "ceiling = time of 64 sequences on one GPU / time of 8 sequences on one GPU (no communication: the shards are independent)"
        out["config5"]["eight_per_gpu"] = eight
        del b5e
    except Exception as e:   # the extras never take the headline down
        out["config5"] = {"error": str(e)}

    # "one non-uniform parameter set, e.g. a fit from --train, as a second case" -- uniform defaults tie every
    # Viterbi choice, a fitted set does not.  Baum-Welch (fitter.py = src/fitter.cpp) on 16 short synthetic pairs, then the
    # headline machine under the fitted parameters on config 4's shape.
    try:
        from machineboss_amd.fitter import MachineFitter
        from machineboss_amd.seqpair import SeqPair
        m4 = Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", "psw2dna.json"))
        train = []
        for k in range(16):
            x, y = synth_tokens(1000 * 4 + 9000 + k, 40, 130, em4.nInTok, em4.nOutTok)
            train.append(SeqPair(em4.inputTokenizer.detokenize(x), em4.outputTokenizer.detokenize(y), "in%d" % k, "out%d" % k))
        t0 = time.perf_counter()
        fitter = MachineFitter(m4)
        params = fitter.fit(train)
        tfit = time.perf_counter() - t0
        allp = dict(m4.funcs); allp.update(params)
        emn = EvaluatedMachine.fromMachine(m4, allp)
        dump_kernels_as("psw2dna_fitted")
        dmn = capi.DeviceMachine(emn)
        bn = capi.DeviceBatch(dmn, *synth_batch(4, 32, 487, 10000, emn.nInTok, emn.nOutTok))
        cellsn = bn.cells()
        lln, tfn = timed(lambda: bn.forward(capi.MB_MATERIALISE), 1, label="nonuniform.forward_materialised"); devn = capi.last_device_ms()
        (vn, offn, en), tvn = timed(lambda: bn.viterbi(paths=True), 1, label="nonuniform.viterbi_with_paths")
        lw = np.asarray(emn.logWeight)
        rescore = float(np.sum(lw[en[offn[0]:offn[1]]]))
        out["nonuniform"] = {"workload": "psw2dna under parameters fitted by Baum-Welch (%d iterations, %.1f s, log-likelihood %.4f -> %.4f on 16 pairs of 40 aa x 130 nt); 32 pairs x 487 aa x 10000 nt"
                                         % (len(fitter.log), tfit, fitter.log[0], fitter.log[-1]),
                             "forward_materialised": round(cellsn / tfn / 1e9, 2), "viterbi_with_paths": round(cellsn / tvn / 1e9, 2), "unit": "Gcells/s (wall clock)",
                             "ms": round(tfn * 1e3, 2), "device_ms": round(devn, 2),
                             "roofline": {"bound": "hbm", "achieved": round(8.0 * cellsn / (devn / 1e3) / 1e9, 1), "peak": hbm_peak, "unit": "GB/s", "frac": round(8.0 * cellsn / (devn / 1e3) / 1e9 / hbm_peak, 4)},
                             "loglike_checksum": float(np.sum(lln)), "viterbi_checksum": float(np.sum(vn)),
                             "path0_rescored_minus_viterbi": rescore - float(vn[0]), "_em": emn, "_dm": dmn}
    except Exception as e:
        out["nonuniform"] = {"error": str(e)}
    return out


def extra_train(capi, np):
    """`boss --train` end to end at N = 1 (VERDICT r5 missing 3; src/fitter.cpp:23-47): wall clock of a full Baum-Welch iteration -- weight
    expressions evaluated, mb_machine_set_weights, E-step, M-step -- on config 3 per GPU (protpsw, 1 024 x 400 x 400 aa; closed-form
    M-step) and on config 5's 5 063-state machine at 64 x 2 kb (weights evaluated, set_weights with the programs re-planned, E-step,
    M-step by BFGS: that machine's weights are sums of products of its 84 parameters, no closed form).  The first iteration carries the
    upload, the kernel specialisation and the tokenisation; `steady` is the mean of the later ones."""
    from machineboss_amd import fitter as F, algebra as A
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.hmmer import HmmerModel
    from machineboss_amd.seqpair import SeqPair
    from machineboss_amd.seqgen import synth_tokens, synth_batch
    P = lambda n: Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", n + ".json"))
    out = {}
    keep = (F.MaxEMIterations, F.MinEMImprovement)
    F.MaxEMIterations, F.MinEMImprovement = 3, -1e300      # a bounded run: four E-steps, three M-steps
    try:
        m = P("protpsw"); nPairs, inLen, outLen = 1024, 400, 400
        em0 = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
        data = []
        for k in range(nPairs):
            x, y = synth_tokens(3000 + k, inLen, outLen, em0.nInTok, em0.nOutTok)
            data.append(SeqPair(em0.inputTokenizer.detokenize(x), em0.outputTokenizer.detokenize(y)))
        fit = F.MachineFitter(m)
        t0 = time.perf_counter(); fit.fit(data); dt = time.perf_counter() - t0
        tl = fit.timing
        mean = lambda key: round(sum(t.get(key, 0.0) for t in tl[1:]) / max(len(tl) - 1, 1), 2)
        steady = {k_: mean(k_) for k_ in ("eval_ms", "set_weights_ms", "estep_ms", "estep_device_ms", "mstep_ms")}
        steady["iteration_ms"] = round(sum(steady[k_] for k_ in ("eval_ms", "set_weights_ms", "estep_ms", "mstep_ms")), 2)
        cells = nPairs * (inLen + 1) * (outLen + 1) * em0.nStates
        out["config3"] = {"workload": "protpsw: %d states, %d transitions, %d pairs x %d x %d aa" % (em0.nStates, em0.nTransitions, nPairs, inLen, outLen), "iterations": len(tl), "total_s": round(dt, 2),
                          "first_iteration_ms": {k_: round(v, 2) for k_, v in tl[0].items()}, "steady": steady,
                          "lattice_gcells_per_s_end_to_end": round(cells / (steady["iteration_ms"] / 1e3) / 1e9, 2), "loglike": [round(x, 4) for x in fit.log]}
    finally:
        F.MaxEMIterations, F.MinEMImprovement = keep
    # config 5 at 64 x 2 kb: the device-facing part of an iteration, three times with slightly different parameters
    m5 = A.composeLeftToRight([HmmerModel.fromFile(os.path.join(ROOT, "tests", "golden", "hmmer", "fn3.hmm")).truncated(20).machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
    params = m5.getParamDefs(True)
    rows = []
    dm5 = b5 = None
    for it in range(4):
        t0 = time.perf_counter()
        ev = EvaluatedMachine.fromMachine(m5, params) if dm5 is None else ev.reweighted(m5, params)      # (as fitter.py does: the weight expressions through a program compiled once)
        t_eval = time.perf_counter() - t0; t0 = time.perf_counter()
        if dm5 is None:
            dm5 = capi.DeviceMachine(ev)
            b5 = capi.DeviceBatch(dm5, *synth_batch(5, 64, 0, 2000, ev.nInTok, ev.nOutTok))
        else:
            dm5.set_weights(ev.logWeight)
        t_set = time.perf_counter() - t0; t0 = time.perf_counter()
        cnt, s, _ = b5.counts()
        t_e = time.perf_counter() - t0
        rows.append({"eval_ms": t_eval * 1e3, "set_weights_ms": t_set * 1e3, "estep_ms": t_e * 1e3, "estep_device_ms": capi.last_device_ms(), "loglike": float(s)})
        # the M-step (src/counts.cpp:117-295): no closed form here (1 905 of the 14 691 weights are sums of products), so BFGS over the
        # reference's transformed parameters, objective and gradient through the compiled program (fitter.MachineObjective)
        t0 = time.perf_counter()
        class _Counts: pass
        cobj = _Counts(); cobj._flat = np.asarray(cnt, np.float64)
        params = F.MachineObjective(m5, cobj, F.Constraints(), {}).optimize(params)
        rows[-1]["mstep_ms"] = (time.perf_counter() - t0) * 1e3
    mean5 = lambda key: round(sum(r[key] for r in rows[1:]) / 3.0, 2)
    steady5 = {k_: mean5(k_) for k_ in ("eval_ms", "set_weights_ms", "estep_ms", "estep_device_ms", "mstep_ms")}
    steady5["device_facing_ms"] = round(steady5["eval_ms"] + steady5["set_weights_ms"] + steady5["estep_ms"], 2)
    steady5["iteration_ms"] = round(steady5["device_facing_ms"] + steady5["mstep_ms"], 2)
    cells5 = 64 * 2001 * ev.nStates
    out["config5_2kb"] = {"workload": "fn3 (20 nodes) . simple_introns . translate . dnapsw: %d states, %d transitions, 64 sequences x 2000 nt" % (ev.nStates, ev.nTransitions),
                          "first_iteration_ms": {k_: round(v, 2) for k_, v in rows[0].items() if k_ != "loglike"}, "steady": steady5,
                          "lattice_gcells_per_s_device_facing": round(cells5 / (steady5["device_facing_ms"] / 1e3) / 1e9, 2),
                          "note": "set_weights only marks the programs stale (0.1 ms): the relaxation, the plan of the merged schedule and the upload are paid inside the E-step that follows (estep_ms - 18.8 ms of a plain E-step = the re-plan)",
                          "loglike": [round(r["loglike"], 4) for r in rows],
                          "mstep": "BFGS on the host (scipy) over the reference's transformed parameterisation; objective and gradient through ONE compiled program of the machine's 14 691 weight expressions (2 ms per evaluation; the scalar way 0.18 s per value)"}
    return out


def extra_dropin(capi, np, quick=False):
    """The boundary as the REFERENCE'S OWN CALLERS drive it: tests/cxx/dropin.cpp runs the `--loglike` loop of
    target/boss.cpp:796-800 and the `--viterbi / --align` loop of :826-833 exactly as written there -- one matrix object per pair --
    through machineboss_amd/cxx/mb_dp.hpp, (a) unchanged, (b) with the ONE line INTEGRATION.md 2b adds in front of each loop
    (MachineBossHIP::prefetch), (c) next to the batch C-ABI.  A child process (C++): this process's cached workspaces are released first.
    Host buffers in, host results out: these rates include tokenising, H2D, D2H and building MachinePath objects."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "cxx"))
    import casefile
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_tokens
    capi.release_workspace()
    out = {"what": "pairs/s and G cells/s THROUGH the drop-in C++ classes for loops shaped like target/boss.cpp:796-800 (RollingOutputForwardMatrix per pair) and :826-833 (ViterbiMatrix per pair: logLike() + path(machine)); matrix_fills = fp64 matrices that crossed PCIe (0: none)"}
    with tempfile.TemporaryDirectory() as tmp:
        exe = casefile.build_exe(tmp, "dropin", "-O2")
        for key, preset, cfg, n, il, ol, reps in (("config2", "dnapsw", 2, 64 if quick else 1024, 1000, 1000, 2), ("config4", "psw2dna", 4, 2 if quick else 8, 487, 10000, 2)):
            em = casefile.file_weights(EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", preset + ".json")), None, useDefaults=True))
            pairs = [synth_tokens(1000 * cfg + k, il, ol, em.nInTok, em.nOutTok) for k in range(n)]
            case = os.path.join(tmp, key + ".txt")
            casefile.write_case(case, em, ["s%d" % s for s in range(em.nStates)], pairs)
            env = {k: v for k, v in os.environ.items() if k != "MB_ROLLING_MIN_PAIRS"}
            r = subprocess.run([exe, case, "time", str(reps)], capture_output=True, text=True, env=env, timeout=1500)
            rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or len(rows) != 3:
                out[key] = {"error": (r.stdout + r.stderr)[-500:]}
                continue
            blk = {"workload": "%s, %d pairs x %d x %d through the C++ classes (mock of the reference's host types)" % (preset, n, il, ol)}
            for row in rows:
                blk[row["variant"]] = {"loglike_pairs_per_s": round(n / row["loglike_s"], 1), "loglike_gcells": row["loglike_gcells"],
                                       "align_pairs_per_s": round(n / row["align_s"], 1), "align_gcells": row["align_gcells"], "matrix_fills": row["matrix_fills"],
                                       "ll_sum": row["ll_sum"], "vit_sum": row["vit_sum"], "path_transitions": row["path_transitions"],
                                       "loglike_s": row["loglike_s"], "align_s": row["align_s"], "inside_prefetch_s": [row["prefetch_loglike_s"], row["prefetch_align_s"]],
                                       "inside_path_objects_s": row["path_objects_s"]}
            a, b, c = (blk[k] for k in ("unchanged_loop", "unchanged_loop_prefetch", "batch_c_abi"))
            # unchanged loop + the one added line against the batch C-ABI fed from the same SeqPairList (tokenising included on both sides);
            # for --align the loop additionally builds the reference's MachinePath objects (std::list<MachineTransition>: host work of the
            # reference's own types that no batch caller of edge ids does) -- quoted with and without that share
            blk["prefetch_over_batch_time"] = {"loglike": round(b["loglike_s"] / c["loglike_s"], 3), "align": round(b["align_s"] / c["align_s"], 3),
                                               "align_without_path_objects": round((b["align_s"] - b["inside_path_objects_s"]) / c["align_s"], 3),
                                               "added_line_alone": [round(b["inside_prefetch_s"][0] / c["loglike_s"], 3), round(b["inside_prefetch_s"][1] / c["align_s"], 3)],
                                               "note": "loop time outside the added line is host work of the reference's own types per pair: eval.canTokenize (two std::map look-ups per symbol), MachinePath = std::list<MachineTransition> built and destroyed (config 2: 3 734 transitions per pair)"}
            blk["same_results"] = bool(a["vit_sum"] == b["vit_sum"] == c["vit_sum"] and a["path_transitions"] == b["path_transitions"] == c["path_transitions"]
                                       and abs(a["ll_sum"] - b["ll_sum"]) <= 1e-8 * abs(b["ll_sum"]) and abs(c["ll_sum"] - b["ll_sum"]) <= 1e-8 * abs(b["ll_sum"]))
            out[key] = blk
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import numpy as np
    # The ranks (machineboss_amd/shard.py, RankGroup): ONE HIP runtime per process -- rendezvous, barriers and the max over ranks
    # go over gloo on the host (torch never touches the GPU), the one data-path collective (E-step counts) over RCCL through the
    # C-ABI on the library's own runtime and stream.  MB_BENCH_BACKEND: "rccl" (default), "nccl" (torch.distributed's NCCL
    # backend = torch's RCCL on torch's runtime, the round-3 route), "gloo" + MB_BENCH_SHARE_DEVICE=1 (every rank on GPU 0, all
    # collectives on the host: the dry run of the N > 1 path on a one-GPU box, tests/test_gpu_parity.py).
    # MB_BENCH_FORCE_COMM=1: a one-rank RCCL communicator at N = 1 (bootstrap + collective exercised on one GPU).
    from machineboss_amd.shard import RankGroup
    grp = RankGroup.from_env(backend=os.environ.get("MB_BENCH_BACKEND"), force=os.environ.get("MB_BENCH_FORCE_COMM") == "1",
                             share_device=os.environ.get("MB_BENCH_SHARE_DEVICE") == "1")
    backend = grp.backend if grp else "none"
    if grp:
        local_rank = grp.local_rank

    from machineboss_amd import capi
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    from machineboss_amd.seqgen import synth_batch, synth_tokens
    from machineboss_amd.shard import shard_range, lpt_assign

    capi.set_device(local_rank)
    global DUMP_DIR
    if rank == 0 and world == 1:
        import tempfile
        DUMP_DIR = tempfile.mkdtemp(prefix="mb_bench_kernels_")
        dump_kernels_as(args.preset)
    if args.dropin_only:
        print(json.dumps(extra_dropin(capi, np, args.quick)))
        return
    m = Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", args.preset + ".json"))
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    dm = capi.DeviceMachine(em)

    if args.scaling == "weak":
        # every rank gets `pairs` pairs; rank r takes pairs [r*pairs, (r+1)*pairs) of the global list
        total_pairs = args.pairs * world
        first, count = shard_range(total_pairs, world, rank)
        mine = list(range(first, first + count))
        inTok, inOff, outTok, outOff = synth_batch(4, count, args.inlen, args.outlen, em.nInTok, em.nOutTok, first=first)
    else:
        # the stated batch (256 pairs) split over the ranks: longest-processing-time-first by DP cell count
        total_pairs = args.pairs
        cells = [(args.inlen + 1) * (args.outlen + 1) * em.nStates] * total_pairs
        mine = lpt_assign(cells, world)[rank]
        parts = [synth_tokens(1000 * 4 + k, args.inlen, args.outlen, em.nInTok, em.nOutTok) for k in mine]
        inOff = np.zeros(len(mine) + 1, np.int64); outOff = np.zeros(len(mine) + 1, np.int64)
        inOff[1:] = np.cumsum([len(a) for a, _ in parts]); outOff[1:] = np.cumsum([len(b) for _, b in parts])
        inTok = np.concatenate([a for a, _ in parts]) if parts else np.zeros(0, np.int32)
        outTok = np.concatenate([b for _, b in parts]) if parts else np.zeros(0, np.int32)
    batch = capi.DeviceBatch(dm, inTok, inOff, outTok, outOff)   # tokens now resident in HBM
    cells_rank = batch.cells()
    flags = capi.MB_MATERIALISE if args.mode == "materialise" else capi.MB_ROLLING

    def sync():
        capi.synchronize()          # (hipDeviceSynchronize on the library's runtime: what torch.cuda.synchronize() is on torch's)
        if grp:
            grp.barrier()
        capi.synchronize()

    ll = None
    for _ in range(args.warmup):
        ll = batch.forward(flags)
    sync()
    dev_ms = 0.0
    launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ll = batch.forward(flags)
        dev_ms += capi.last_device_ms()
        launches += capi.last_launch_count()
    dt_rank = time.perf_counter() - t0      # this rank's own clock (before the closing barrier): reported per rank at N > 1
    sync()
    dt = time.perf_counter() - t0
    kernel = capi.last_kernel_name()
    total_cells = cells_rank
    if world > 1:
        dt = grp.all_reduce_float(dt, "max")
        total_cells = grp.all_reduce_float(float(cells_rank), "sum")
    value = total_cells * args.steps / dt / 1e9

    extra = {}
    if world > 1:
        # every rank's share and clock: an imbalance (ragged shard, a slow device) shows as one rank's seconds standing out
        # ... and its log-likelihood checksum beside the committed per-pair values of the same synthetic pairs (profiles/<tag>_loglike_per_pair.json,
        # written on one GPU by scripts/loglike_reference.py): a rank that computed on the wrong device, the wrong shard or with a broken
        # communicator shows here, on first contact with a multi-GPU node
        ref = recorded(PROFILE_TAG + "_loglike_per_pair.json")
        ref_ok = bool(ref) and ref.get("preset") == args.preset and ref.get("inlen") == args.inlen and ref.get("outlen") == args.outlen and (max(mine) if mine else 0) < len(ref.get("loglike", []))
        ref_sum = float(sum(ref["loglike"][k] for k in mine)) if ref_ok else float("nan")
        rows = sorted(grp.all_gather_floats([float(rank), float(cells_rank), dt_rank, dev_ms / 1e3, float(np.sum(ll)), ref_sum, float(len(mine))]))
        extra["per_rank"] = [{"rank": int(r[0]), "cells_per_step": int(r[1]), "seconds": round(r[2], 4), "device_seconds": round(r[3], 4),
                              "gcells_per_s": round(r[1] * args.steps / max(r[2], 1e-12) / 1e9, 2), "pairs": int(r[6]), "loglike_checksum": r[4],
                              "loglike_reference": (None if r[5] != r[5] else r[5])} for r in rows]
        cells_expected = total_pairs * (args.inlen + 1) * (args.outlen + 1) * em.nStates
        checks = {"n_ranks_seen": len(rows), "ranks_distinct": len({int(r[0]) for r in rows}) == world,
                  "cells_all_ranks": int(sum(r[1] for r in rows)), "cells_expected": int(cells_expected), "pairs_all_ranks": int(sum(r[6] for r in rows)), "pairs_expected": int(total_pairs),
                  "loglike_checksum_all_ranks": float(sum(r[4] for r in rows)),
                  "loglike_reference": ("profiles/%s_loglike_per_pair.json" % PROFILE_TAG) if all(r[5] == r[5] for r in rows) else None,
                  "loglike_max_rel_dev_from_reference": (max(abs(r[4] - r[5]) / max(abs(r[5]), 1e-300) for r in rows) if all(r[5] == r[5] for r in rows) else None)}
        checks["ok"] = bool(checks["n_ranks_seen"] == world and checks["ranks_distinct"] and checks["cells_all_ranks"] == checks["cells_expected"] and checks["pairs_all_ranks"] == checks["pairs_expected"]
                            and all(np.isfinite(r[4]) for r in rows) and (checks["loglike_max_rel_dev_from_reference"] is None or checks["loglike_max_rel_dev_from_reference"] <= 1e-8))
        extra["checks"] = checks
        slow = max(rows, key=lambda r: r[2])
        extra["slowest_rank"] = {"rank": int(slow[0]), "seconds": round(slow[2], 4), "over_mean": round(slow[2] / (sum(r[2] for r in rows) / world), 4)}
        # what the sharding can deliver, per BASELINE config (pairs are independent units; nothing but --train's counts is exchanged)
        extra["expected_scaling"] = {
            "config 2-4 (batches of pairs, this line)": "weak scaling 1.0 per GPU by construction: every rank fills its own pairs, no data-path collective; strong scaling of 256 pairs over 8 GPUs leaves 32 pairs per GPU, whose tile wavefront still fills 256 CUs (672 live tiles per launch)",
            "config 3 (--train)": "one all-reduce of nTransitions + 1 doubles (3.6 KB) per EM iteration: latency only",
            "config 5 (64 sequences x 50 kb over 8 GPUs)": "STRONG scaling ceiling 1.0-1.3x for the 5 063-state machine (measured on one GPU, extra.config5.eight_per_gpu of the N = 1 line: 64 sequences against 8; round 6 with the generated sweeps: Forward 1.27, Viterbi 1.06, --align 1.04): a one-tape lattice is serial along its columns, a sequence is k <= 8 workgroups and its period a latency chain of 5 rounds (DESIGN.md 4.4), so 8 sequences per GPU take almost as long as 64 on one GPU; only a batch of more sequences than CUs scales.  The whole fn3 profile (21 761 states, ring beyond one CU's LDS) does scale with workgroups per sequence: 64 -> 16 -> 4 sequences x 3 kb take 30 -> 18 -> 15 ms (Viterbi fill)"}
    if grp and not args.no_extra:
        # the ONE collective of the path (--train): E-step on this rank's shard of config 3, then the all-reduce of
        # nTransitions + 1 doubles over RCCL (xGMI)
        mp = Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", "protpsw.json"))
        emp = EvaluatedMachine.fromMachine(mp, None, useDefaults=True)
        dmp = capi.DeviceMachine(emp)
        per = 1024
        bp = capi.DeviceBatch(dmp, *synth_batch(3, per, 400, 400, emp.nInTok, emp.nOutTok, first=rank * per))
        bp.counts()
        seen = grp.all_reduce_float(1.0, "sum")
        sync(); t1 = time.perf_counter()
        its = 3
        for _ in range(its):
            cnt, s, _ = bp.counts()
            te = time.perf_counter()
            cnt, s = grp.allreduce_counts(cnt, s)
            tr = time.perf_counter() - te
        sync(); d1 = (time.perf_counter() - t1) / its
        nsym = float(cnt[np.asarray(emp.inTok) != 0].sum())
        extra["em_iteration"] = {"workload": "config 3: protpsw --train E-step, %d x 400 x 400 aa per GPU + all-reduce of %d doubles (backend %s)" % (per, emp.nTransitions + 1, {"rccl": "RCCL through the C-ABI", "nccl": "torch nccl = RCCL"}.get(backend, backend)),
                                 "ms_per_iteration": round(d1 * 1e3, 3), "allreduce_ms": round(tr * 1e3, 3), "n_ranks_seen": int(round(seen)),
                                 "value": round(world * bp.cells() / d1 / 1e9, 2), "unit": "G lattice-cells/s over all ranks",
                                 "symbol_count_invariant": nsym / (world * per * 400)}
    if world == 1 and rank == 0 and not args.no_extra and not args.extra_em_only:
        other = capi.MB_ROLLING if flags == capi.MB_MATERIALISE else capi.MB_MATERIALISE
        _, d1 = timed(lambda: batch.forward(other), label="headline.other_mode")
        extra["rolling_gcells_per_gpu" if other == capi.MB_ROLLING else "materialised_gcells_per_gpu"] = round(cells_rank / d1 / 1e9, 3)
        if other == capi.MB_ROLLING:
            # the rolling mode is priced against the transcendental issue rate, not HBM
            ops = capi.sweep_ops(dm)
            tr = (ops["exp_per_cell"] + ops["log_per_cell"]) * cells_rank / d1
            extra["rolling_rate"] = {"exp_per_cell": round(ops["exp_per_cell"], 3), "log_per_cell": round(ops["log_per_cell"], 3), "family": ops["family"],
                                     "achieved_transcendental_per_s": round(tr / 1e12, 3), "peak": TRANSCENDENTAL_PEAK_T, "unit": "T v_exp/v_log per s",
                                     "frac": round(tr / 1e12 / TRANSCENDENTAL_PEAK_T, 4),
                                     "issue": valu_issue(["psw2dna.sum.roll.fwd.clos.hip"], cells_rank / d1),
                                     "note": "quarter-rate fp32 transcendentals: 157 TFLOP/s fp32 / 2 / 4 (MI355X_MICROARCH.md); the sweep is bound by total vector issue (fp64 add / max at half rate), of which these are a part"}
            extra["rolling_note"] = "boss --loglike mode: no matrix in HBM, bound by vector instruction issue (fp64 add/max, v_exp_f32/v_log_f32), not by HBM; the HBM fraction is not meaningful for it"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg runs at N = 1 only (the other ranks would idle behind it)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle   # checker / baseline only: never on the product path
        om = oracle.OracleMachine(em)
        # one pair per host core, all cores at once: the C restatement is single-threaded like the reference, pairs are
        # independent, and ctypes releases the GIL during the call.  Sample sized for ~10-20 s of wall time.
        cores = host_cores()
        # calibrate: a container may show more CPUs than it is allowed to run; use as many threads as actually scale
        probe = [synth_tokens(3000 + k, args.inlen, 150, em.nInTok, em.nOutTok) for k in range(cores)]
        om.loglike(*probe[0])
        p1, pn = 1e9, 1e9
        for _ in range(2):      # best of two: the probe is short and the host is shared
            tp = time.perf_counter(); om.loglike(*probe[0]); p1 = min(p1, time.perf_counter() - tp)
            with ThreadPoolExecutor(max_workers=cores) as ex:
                tp = time.perf_counter(); list(ex.map(lambda xy: om.loglike(*xy), probe)); pn = min(pn, time.perf_counter() - tp)
        cores = max(1, min(cores, int(round(cores * p1 / pn))))
        # the sample is the workload's own shape (487 aa x 10 kb) when that fits ~25 s of CPU wall time, else shortened DNA
        per_core = (args.inlen + 1) * 151 * em.nStates / p1            # cells/s of one core
        sample_out = int(min(args.outlen, max(1500, 25.0 * per_core / ((args.inlen + 1) * em.nStates))))
        samples = [synth_tokens(4000 + k, args.inlen, sample_out, em.nInTok, em.nOutTok) for k in range(cores)]
        om.loglike(samples[0][0][:50], samples[0][1][:200])
        with ThreadPoolExecutor(max_workers=cores) as ex:
            t2 = time.perf_counter(); refs = list(ex.map(lambda xy: om.loglike(*xy), samples)); d2 = time.perf_counter() - t2
        t2 = time.perf_counter(); ref1 = om.loglike(samples[0][0], samples[0][1][:1500]); d1 = time.perf_counter() - t2      # single-core rate
        sample_cells = (args.inlen + 1) * (sample_out + 1) * em.nStates
        b1 = capi.DeviceBatch.from_pairs(dm, samples[:2])
        got = b1.forward(flags)
        assert all(abs(g - r) <= 1e-4 * abs(r) for g, r in zip(got, refs[:2])) and np.isfinite(ref1)   # same sample through the GPU path: parity at bench scale
        single = (args.inlen + 1) * 1501 * em.nStates / d1 / 1e9
        cpu = {"value": round(cores * sample_cells / d2 / 1e9, 5), "unit": "Gcells/s", "cores": cores, "kind": "port",
               "single_core_value": round(single, 5),
               # the REFERENCE's own rate is a recorded constant (its sources need GSL / Boost and cannot be built or travel):
               "reference_per_core": {"value": 0.0116, "unit": "Gcells/s", "what": "the reference's src/*.cpp (RollingOutputForwardMatrix, -O3), psw2dna 100 aa x 100 nt, one core of a 2.1 GHz Xeon",
                                      "source": "BASELINE.md section 2 (survey probe), recorded -- not measured in this run"},
               "socket_extrapolation": {"cores": 64, "port_gcells": round(single * 64, 3), "reference_gcells": round(0.0116 * 64, 3),
                                        "note": "one 64-core socket, perfect scaling over independent pairs assumed for both; the north star's '>= 50x single-socket CPU' is read against these"},
               "sample": "%d pairs (one per host core, concurrently) of %d aa x %d nt on %s (%.1f s wall), RollingOutputForwardMatrix restatement oracle/mb_oracle.c, table logsumexp"
                         % (cores, args.inlen, sample_out, args.preset, d2)}

    if rank == 0 and world == 1 and not args.no_extra and not args.extra_em_only:
        extra.update(extra_single_gpu(capi, np, HBM_PEAK_GBS))
        try:
            extra["train"] = extra_train(capi, np)
        except Exception as e:
            extra["train"] = {"error": str(e)}
        try:
            extra["dropin"] = extra_dropin(capi, np)
        except Exception as e:
            extra["dropin"] = {"error": str(e)}
        nu = extra.get("nonuniform", {})
        emn, dmn = nu.pop("_em", None), nu.pop("_dm", None)
        if emn is not None and not args.no_cpu:
            # CPU leg, checker only: the fitted machine through the oracle on two short pairs (Forward within 1e-4, Viterbi
            # score and path bit for bit -- with non-uniform parameters the path is a genuine arg-max, not a chain of ties)
            from oracle import oracle
            omn = oracle.OracleMachine(emn)
            small = [synth_tokens(1000 * 4 + 9500 + k, 35 + 5 * k, 110 + 20 * k, emn.nInTok, emn.nOutTok) for k in range(2)]
            bs = capi.DeviceBatch.from_pairs(dmn, small)
            gl = bs.forward(capi.MB_MATERIALISE); gv, go, ge = bs.viterbi(paths=True)
            ok = True; worst = 0.0
            for k, (x, y) in enumerate(small):
                r = omn.loglike(x, y); V = omn.viterbi(x, y)
                worst = max(worst, abs(gl[k] - r) / abs(r))
                ok = ok and abs(gl[k] - r) <= 1e-4 * abs(r) and gv[k] == V[-1, -1, -1] and np.array_equal(ge[go[k]:go[k + 1]], omn.traceback(x, y, V))
            nu["oracle_check"] = {"pairs": [[len(x), len(y)] for x, y in small], "forward_max_rel_err": worst, "viterbi_and_paths_bit_exact": bool(ok), "ok": bool(ok)}
            assert ok, "non-uniform parameter case disagrees with the oracle"

    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes per launch / average launch duration (HIP events on the
        # library stream around the launch sequence; launches are back to back, gaps < 1 us in the rocprof trace)
        ach = BYTES_PER_CELL * cells_rank * args.steps / (dev_ms / 1e3) / 1e9 if (dev_ms > 0 and flags == capi.MB_MATERIALISE) else 0.0
        traffic = None
        traffic_src = None
        pmc = recorded(PROFILE_TAG + "_pmc_hbm.json")      # HBM bytes per launch from the committed PMC passes: a recorded constant, valid for the default workload and the kernel it was measured on
        if pmc and pmc.get("kernel") == kernel and pmc.get("cells_per_step") == cells_rank and flags == capi.MB_MATERIALISE:
            if kernels_unchanged("traffic:headline", ["psw2dna.sum.mat.fwd.clos"]):
                traffic = round(pmc["hbm_bytes_per_cell"] * cells_rank * args.steps / max(launches, 1))
                traffic_src = "profiles/%s_pmc_hbm.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command; recorded -- kernel source verified unchanged by hash)" % PROFILE_TAG
            else:
                traffic_src = REFUSED.get("traffic:headline")
        elif flags == capi.MB_MATERIALISE:
            traffic_src = "no PMC passes recorded for this workload (profiles/%s_pmc_hbm.json)" % PROFILE_TAG
        out = {
            "metric": "Giga DP-cells/sec (Forward) on composed protpsw machine",
            "value": round(value, 3), "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4a: %s (%d states, %d transitions), %d pairs%s x %d aa x %d nt, Forward %s, --use-defaults params; inputs resident in HBM when the clock starts (tokens: %.1f MB of H2D per GPU, not timed), log-likelihoods copied back inside the timed region"
                                   % (args.preset, em.nStates, em.nTransitions, args.pairs, "/GPU" if args.scaling == "weak" else " in total", args.inlen, args.outlen, args.mode, (len(inTok) + len(outTok)) * 4 / 1e6),
                       "parallelism": "pairs sharded over %d GPU(s), no data-path collective" % world,
                       "cells_per_gpu_per_step": int(cells_rank),
                       "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MB_")}},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL if flags == capi.MB_MATERIALISE else 0,
                         "algorithmic_bytes_per_launch": round(BYTES_PER_CELL * cells_rank * args.steps / max(launches, 1)) if flags == capi.MB_MATERIALISE else 0,
                         "launches_per_step": launches // max(args.steps, 1),
                         "avg_launch_us": round(dev_ms * 1e3 / max(launches, 1), 2),
                         "device_ms_per_step": round(dev_ms / args.steps, 3),
                         "issue": valu_issue(["psw2dna.sum.mat.fwd.clos.hip" if flags == capi.MB_MATERIALISE else "psw2dna.sum.roll.fwd.clos.hip"],
                                             cells_rank * args.steps / max(dev_ms / 1e3, 1e-9)) if args.preset == "psw2dna" else None},
            "cpu_baseline": cpu,
            "loglike_checksum": float(np.sum(ll)),      # this rank's (rank 0); all ranks: extra.checks.loglike_checksum_all_ranks
            "recorded_constants": {"tag": PROFILE_TAG, "refused": dict(REFUSED), "kernel_sha": kernel_sha_now()},
            "extra": extra,
        }
        if extra:
            # wall clock beside device time for every timed mode, and the modes where the two differ
            extra["timings"] = {"unit": "[wall ms, device ms] per call", **TIMINGS}
            out["host_overhead_flags"] = host_overhead_flags()
            out["alloc_stats"] = capi.alloc_stats()
        if args.write_kernel_sha:
            with open(args.write_kernel_sha, "w") as fh:
                json.dump({"written_by": "python bench.py --write-kernel-sha (the round's profile run)", "kernels": kernel_sha_now()}, fh, indent=1, sort_keys=True)
        # the figures of `extra` that other documents quote, once more at the END of the line (a reader that keeps only the tail of a
        # long line still sees them): mode -> [rate, HBM-roofline fraction or None]
        if extra:
            def _rf(k, rate_key="value", roof="roofline"):
                v = extra.get(k) or {}
                return [v.get(rate_key), (v.get(roof) or {}).get("frac")] if rate_key in v else None
            out["summary"] = {"headline": [out["value"], out["roofline"]["frac"]], "rolling_gcells": extra.get("rolling_gcells_per_gpu"),
                              "counts4": _rf("counts4"), "viterbi4_with_paths": _rf("viterbi4"), "counts_config3": _rf("counts"), "forward_config3": _rf("forward_config3"),
                              "config4b_forward_materialised": _rf("config4b", "forward_materialised"), "config4b_counts": _rf("config4b", "counts_lattice", "roofline_counts"),
                              "nonuniform_forward_materialised": [(extra.get("nonuniform") or {}).get("forward_materialised"), ((extra.get("nonuniform") or {}).get("roofline") or {}).get("frac")],
                              "host_overhead_flags": out.get("host_overhead_flags"),
                              "train_iteration_ms_config3_config5_2kb": [((extra.get("train") or {}).get(k_) or {}).get("steady", {}).get("iteration_ms") for k_ in ("config3",)] + [(((extra.get("train") or {}).get("config5_2kb") or {}).get("steady") or {}).get("iteration_ms")],
                              "config5_50kb_forward_viterbi_withpaths": [((extra.get("config5") or {}).get("full_size") or {}).get(k) for k in ("forward_rolling", "viterbi_fill", "viterbi_with_paths")],
                              "unit": "G cells/s (counts: G lattice-cells/s), fraction of 8 TB/s at the mode's algorithmic bytes"}
        print(json.dumps(out))
        sys.stdout.flush()
    failed = rank == 0 and world > 1 and not (extra.get("checks") or {}).get("ok", False)
    if grp and not args.no_extra and world > 1 and rank == 0 and (extra.get("em_iteration") or {}).get("n_ranks_seen") != world:
        failed = True
    if grp:
        grp.close()
    if failed:
        sys.stderr.write("bench.py: the multi-rank consistency checks FAILED (extra.checks / em_iteration.n_ranks_seen): the line above is not a valid measurement\n")
        sys.exit(4)


if __name__ == "__main__":
    main()

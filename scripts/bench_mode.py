"""One mode of the hot path on its BASELINE config, a fixed number of calls -- the command scripts/profile_modes.sh puts
under rocprofv3 (kernel trace, then separate WRITE_SIZE / FETCH_SIZE passes).
usage: python scripts/bench_mode.py <counts|viterbi|forward3|forward2|counts4|viterbi4|forward4b|c4b_counts> [reps=3]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
mode = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
preset, cfg, n, il, ol = {"counts": ("protpsw", 3, 1024, 400, 400), "forward3": ("protpsw", 3, 1024, 400, 400),
                          "viterbi": ("dnapsw", 2, 1024, 1000, 1000), "forward2": ("dnapsw", 2, 1024, 1000, 1000),
                          "counts4": ("psw2dna", 4, 21, 487, 10000), "viterbi4": ("psw2dna", 4, 64, 487, 10000),
                          "forward4b": ("c4b", 4, 64, 487, 10000), "c4b_counts": ("c4b", 4, 24, 487, 10000)}[mode]   # config 4's own shape: one chunk of Backward matrices / 64 pairs of traceback bytes
if preset == "c4b":      # config 4 read literally: protpsw . translate . dnapsw, composed here (482 states); DNA over {A,C,G}
    from machineboss_amd import algebra
    m = algebra.config4bMachine("tests/golden/preset")
else:
    m = Machine.fromFile("tests/golden/preset/%s.json" % preset)
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(cfg, n, il, ol, em.nInTok, 3 if preset == "c4b" else em.nOutTok))
fn = {"counts": lambda: b.counts(), "counts4": lambda: b.counts(), "c4b_counts": lambda: b.counts(), "viterbi4": lambda: b.viterbi(paths=True), "viterbi": lambda: b.viterbi(paths=True), "forward3": lambda: b.forward(capi.MB_MATERIALISE),
      "forward2": lambda: b.forward(capi.MB_MATERIALISE), "forward4b": lambda: b.forward(capi.MB_MATERIALISE)}[mode]
held = []; dev = []
t0 = time.perf_counter()
for _ in range(reps):
    held.append(fn()); dev.append(capi.last_device_ms())
dt = (time.perf_counter() - t0)
print(json.dumps({"mode": mode, "workload": "%s %d x %d x %d" % (preset, n, il, ol), "calls": reps, "cells_per_call": b.cells(),
                  "device_ms": dev, "wall_ms_total_incl_first_call": dt * 1e3, "kernel": capi.last_kernel_name(), "jit": capi.jit_stats()}))

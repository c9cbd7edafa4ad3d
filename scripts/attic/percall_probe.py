"""Per-call cost of ONE pair through the batch C-ABI (what the lazy ViterbiMatrix / RollingOutputForwardMatrix of the drop-in classes
pay per pair in an unchanged boss.cpp loop): batch create / rolling Forward / Viterbi with path / destroy, wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
capi.set_device(0)
for preset, il, ol, reps in (("dnapsw", 1000, 1000, 100), ("protpsw", 50, 50, 100), ("psw2dna", 487, 10000, 5)):
    if len(sys.argv) > 1 and preset not in sys.argv[1:]:
        continue
    em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", preset + ".json")), None, useDefaults=True)
    dm = capi.DeviceMachine(em)
    pairs = [synth_tokens(7 + k, il, ol, em.nInTok, em.nOutTok) for k in range(4)]
    T = {"create": 0.0, "forward": 0.0, "viterbi": 0.0, "close": 0.0}; dev = {"forward": 0.0, "viterbi": 0.0}
    for k in range(reps + 2):
        x, y = pairs[k % 4]
        t0 = time.perf_counter(); b = capi.DeviceBatch.from_pairs(dm, [(x, y)]); t1 = time.perf_counter()
        b.forward(capi.MB_ROLLING); t2 = time.perf_counter(); d1 = capi.last_device_ms()
        b.viterbi(paths=True); t3 = time.perf_counter(); d2 = capi.last_device_ms()
        b.close(); t4 = time.perf_counter()
        if k >= 2:
            T["create"] += t1 - t0; T["forward"] += t2 - t1; T["viterbi"] += t3 - t2; T["close"] += t4 - t3; dev["forward"] += d1; dev["viterbi"] += d2
    print(preset, il, ol, {k: round(v / reps * 1e6, 1) for k, v in T.items()}, "us per call; device ms", {k: round(v / reps, 4) for k, v in dev.items()}, flush=True)

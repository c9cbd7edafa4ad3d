"""Host-side program builders (both code generators, every mode, presets and random machines) under AddressSanitizer /
UBSan: scripts/asan_host.sh builds the sanitised library and runs this (CPU only; no GPU sanitizers on this pool)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from machineboss_amd import capi
capi.LIB_PATH = os.environ.get('MBHIP_ASAN_LIB', capi.LIB_PATH)
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from randmachine import random_machine
OUT = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'mbhip_asan_out.hip')
n = 0
for preset in ("dnapsw", "protpsw", "psw2dna"):
    em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(ROOT, 'tests', 'golden', 'preset', preset + '.json')), None, useDefaults=True)
    if em.nStates <= 16:
        for mode in (0, 1, 2, 3):
            for bw in (False, True):
                for env in (False, True):
                    if mode == 3 and bw: continue
                    capi.debug_small_source(em, OUT, mode=mode, backward=bw, materialise=(mode != 3), envelopes=env); n += 1
    else:
        for mode, clos in ((capi.MB_FORWARD, 1), (capi.MB_FORWARD, 2), (capi.MB_FORWARD, 0), (capi.MB_VITERBI, 0), (3, 0)):
            for bw in (False, True):
                if mode != capi.MB_FORWARD and bw: continue
                for G in (1, 2, 4, 8):
                    capi.debug_jit_source(em, OUT, mode=mode, backward=bw, closure=clos, G=G); n += 1
rng = np.random.RandomState(1)
for seed in range(60):
    S = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 17, 33, 64, 100, 257, 300]))
    em = random_machine(S, int(rng.randint(1, 4)), int(rng.randint(1, 4)), seed, density=float(rng.uniform(0.8, 3.0)), silent_density=float(rng.uniform(0.2, 2.0)), allow_inf=(seed % 5 == 0))
    try:
        if S <= 16:
            for mode in (0, 2, 3):
                capi.debug_small_source(em, OUT, mode=mode, materialise=(mode != 3), envelopes=bool(seed % 2)); n += 1
            capi.debug_small_source(em, OUT, mode=0, backward=True); n += 1
        else:
            for mode, clos, bw in ((capi.MB_FORWARD, 1, False), (capi.MB_FORWARD, 3, True), (capi.MB_VITERBI, 0, False), (3, 0, False)):
                capi.debug_jit_source(em, OUT, mode=mode, backward=bw, closure=clos, G=int(rng.choice([1, 2, 4, 8, 16]))); n += 1
    except capi.MbError as e:
        print("refused:", S, str(e)[:80])
# the one-tape family: retimed planner, the cuts (k workgroups per sequence, merged candidates), the generator of mb_wide_jit.cpp (round 6) --
# every plan attempt, traceback codes, the fp64 term; random block machines and a profile composite
from randmachine import random_block_machine
from machineboss_amd import algebra as A
from machineboss_amd.hmmer import HmmerModel
P = lambda name: Machine.fromFile(os.path.join(ROOT, 'tests', 'golden', 'preset', name + '.json'))
h = HmmerModel.fromFile(os.path.join(ROOT, 'tests', 'golden', 'hmmer', 'fn3.hmm')).truncated(3)
machines = [EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P('simple_introns'), P('translate'), P('dnapsw')]), None, useDefaults=True)]
for seed in range(12):
    machines.append(random_block_machine(int(rng.choice([3, 5, 8])), int(rng.choice([6, 12, 25])), 0 if seed % 3 else 3, 3 if seed % 3 else 0, 700 + seed))
for em in machines:
    for k in (1, 2, 4):
        for mode, bw, tb, acc in ((capi.MB_VITERBI, False, False, False), (capi.MB_VITERBI, False, True, False), (capi.MB_FORWARD, False, False, True), (capi.MB_FORWARD, True, False, False)):
            for attempt in ("0", "2", "3"):
                os.environ["MB_WIDE_JIT_ATTEMPT"] = attempt
                try:
                    capi.debug_wide_jit(em, OUT, k=k, lanes=int(rng.choice([0, 64, 256])), mode=mode, backward=bw, tb_codes=tb, acc=acc); n += 1
                except (capi.MbError, RuntimeError) as e:
                    print("refused:", em.nStates, k, str(e)[:80])
                    break
os.environ.pop("MB_WIDE_JIT_ATTEMPT", None)
print("host exercise done:", n, "program builds")

#!/bin/bash
# The HOST side of the library under AddressSanitizer + UBSan, without a device: every source compiled with -fsanitize -fno-gpu-sanitize (the
# kernels are built as usual: GPU sanitizers are not available on this pool), linked into a scratch libmbhip_asan.so, and the planners that
# need no device driven through the debug entry points -- mb_debug_wide_retimed, mb_debug_wide_parts (the cut, the two-transition candidates,
# the lane search), mb_debug_jit_source, mb_debug_small_source -- on the golden machines and random block machines.
# usage: bash scripts/asan_planner.sh   (CPU only, ~3 minutes)
set -u
ROOT=$(pwd)
T=$(mktemp -d)
CS=machineboss_amd/csrc
for s in $(ls $CS | grep -E '\.(hip|cpp)$'); do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -c $CS/$s -o $T/$s.o -I include -I $CS -O1 -g -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -munsafe-fp-atomics \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function -Wno-unused-value 2> $T/$s.err &
done
wait
for s in $(ls $CS | grep -E '\.(hip|cpp)$'); do [ -f $T/$s.o ] || { echo "compile failed: $s"; tail -5 $T/$s.err; exit 1; }; done
/opt/rocm/bin/hipcc -shared -o $T/libmbhip_asan.so $T/*.o --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -lhiprtc 2> $T/link.err || { echo "link failed"; tail -5 $T/link.err; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan*x86_64*.so' | head -1)
echo "sanitizer runtime: $RT; instrumented symbols in the library: $(nm -D $T/libmbhip_asan.so | grep -c __asan_)"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
MBHIP_LIBRARY=$T/libmbhip_asan.so LD_PRELOAD=$RT python3 - > $T/run.out 2> $T/run.err <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from machineboss_amd import capi
assert "asan" in capi.LIB_PATH
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd import algebra as A
from randmachine import random_block_machine, random_machine
tmp = tempfile.mkdtemp()
P = lambda n: Machine.fromFile("tests/golden/preset/%s.json" % n)
n = 0
for nodes in (2, 5):
    h = HmmerModel.fromFile("tests/golden/hmmer/fn3.hmm").truncated(nodes)
    em = EvaluatedMachine.fromMachine(A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")]), None, useDefaults=True)
    for mode, bwd, tb in ((capi.MB_VITERBI, False, True), (capi.MB_FORWARD, False, False), (capi.MB_FORWARD, True, False)):
        capi.debug_wide_retimed(em, tmp + "/r.bin", mode, bwd, tb_codes=tb); n += 1
        for k, lanes in ((2, 0), (4, 256), (7, 64)):
            capi.debug_wide_parts(em, tmp + "/p.bin", k, lanes, mode, bwd, tb_codes=tb); n += 1
for c in range(40):
    em = random_block_machine(2 + c % 9, 3 + (c * 7) % 20, 0 if c % 3 else 2, 3 if c % 3 else 0, 500 + c, density=1.0 + (c % 5) * 0.5, silent_density=0.3 + (c % 4) * 0.6, allow_inf=c % 4 == 0)
    for k, lanes in ((2, 64), (3, 128), (5, 0)):
        try: capi.debug_wide_parts(em, tmp + "/p.bin", k, lanes, capi.MB_VITERBI, False, tb_codes=c % 2 == 0); n += 1
        except capi.MbError: pass
for name in ("psw2dna", "dnapsw", "protpsw"):
    em = EvaluatedMachine.fromMachine(P(name), None, useDefaults=True)
    if em.nStates > 16:
        for mode in (0, 16, 3 + 16): capi.debug_jit_source(em, tmp + "/j.hip", mode=mode, closure=2, G=2); n += 1
    else:
        for mode in (0, 1, 2, 3): capi.debug_small_source(em, tmp + "/s.hip", mode=mode); n += 1
print("planner calls under the sanitizers:", n)
PY
echo "rc $?  $(tail -1 $T/run.out)  sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error' $T/run.err)"
# (On a GPU box the same build cannot run real calls: ROCm's AddressSanitizer runtime intercepts hsa_amd_memory_pool_allocate and needs XNACK,
#  which this pool does not offer -- the launch side is covered by the GPU tests and fuzzers with the product build.)
grep -h -A8 'ERROR: AddressSanitizer\|runtime error' $T/run.err | head -40
tail -3 $T/run.err | cut -c1-300

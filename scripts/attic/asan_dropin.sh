#!/bin/bash
# The C++ shim (mb_dp.hpp: lazy matrices, prefetch store, DeviceBatch) under AddressSanitizer + UBSan on the HOST side: tests/cxx/dropin.cpp
# and tests/cxx/test_glue.cpp built with -fsanitize, run against the product libmbhip.so on the GPU box (the device code is not instrumented:
# GPU sanitizers are not available on this pool).  usage (through gpurun): bash scripts/asan_dropin.sh
set -u
T=$(mktemp -d)
python3 - "$T" <<'PY'
import os, sys
sys.path.insert(0, "tests/cxx"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import casefile
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_tokens
for name, shapes in (("dnapsw", [(45, 52), (3, 80), (70, 9)]), ("psw2dna", [(9, 31), (14, 20)])):
    em = casefile.file_weights(EvaluatedMachine.fromMachine(Machine.fromFile("tests/golden/preset/%s.json" % name), None, useDefaults=True))
    pairs = [synth_tokens(90 + k, il, ol, em.nInTok, em.nOutTok) for k, (il, ol) in enumerate(shapes)]
    casefile.write_case(os.path.join(sys.argv[1], name + ".txt"), em, ["s%d" % s for s in range(em.nStates)], pairs, seed=5)
PY
L=$(pwd)/machineboss_amd
for exe in dropin test_glue; do
  g++ -std=c++14 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -DMB_GLUE_MOCK -I include -I machineboss_amd/cxx -I tests/cxx tests/cxx/$exe.cpp -o $T/$exe -L $L -lmbhip -Wl,-rpath,$L || exit 1
done
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
for name in dnapsw psw2dna; do
  $T/dropin $T/$name.txt check > $T/$name.check.out 2> $T/$name.check.err; echo "dropin check $name: rc $? $(tail -1 $T/$name.check.out) sanitizer lines: $(grep -c 'ERROR: AddressSanitizer\|runtime error' $T/$name.check.err)"
  $T/dropin $T/$name.txt time 1 > $T/$name.time.out 2> $T/$name.time.err; echo "dropin time $name: rc $? $(tail -1 $T/$name.time.out) sanitizer lines: $(grep -c 'ERROR: AddressSanitizer\|runtime error' $T/$name.time.err)"
  $T/test_glue $T/$name.txt > $T/$name.glue.out 2> $T/$name.glue.err; echo "test_glue $name: rc $? $(tail -1 $T/$name.glue.out) sanitizer lines: $(grep -c 'ERROR: AddressSanitizer\|runtime error' $T/$name.glue.err)"
  grep -h -A6 'ERROR: AddressSanitizer\|runtime error' $T/$name.*.err | head -20
done

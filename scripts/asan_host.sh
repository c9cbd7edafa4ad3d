#!/bin/bash
# CPU-only sanitizer run of the library's HOST code (program compilers, code generators): builds libmbhip with
# -fsanitize=address,undefined for the host side only (-fno-gpu-sanitize) under $TMPDIR and runs scripts/asan_host_exercise.py.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/mbhip_asan; mkdir -p "$OUT"; cd "$OUT"
for f in "$ROOT"/machineboss_amd/csrc/*.hip "$ROOT"/machineboss_amd/csrc/*.cpp; do
  /opt/rocm/bin/hipcc -x hip -c "$f" -o "$(basename "$f").o" -I"$ROOT/include" -I"$ROOT/machineboss_amd/csrc" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC \
    -munsafe-fp-atomics -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer &
done; wait
/opt/rocm/bin/hipcc -shared -o libmbhip_asan.so ./*.o --offload-arch=gfx950 -lhiprtc -fsanitize=address,undefined
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
MBHIP_ASAN_LIB=$OUT/libmbhip_asan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT \
  python3 "$ROOT/scripts/asan_host_exercise.py"

"""Dump the run-time specialised kernel of a preset and cross-compile it to gfx950 ISA (no GPU needed).

usage: python scripts/jit_dump.py <preset> [mode fwd|vit|bwd|cnt|tb|cntroll|fwdroll] [G] [outdir]
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
preset = sys.argv[1]; mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"; G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/jit"
os.makedirs(out, exist_ok=True)
m = Machine.fromFile("tests/golden/preset/%s.json" % preset); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
src = os.path.join(out, "%s_%s_G%d.hip" % (preset, mode, G))
capi.debug_jit_source(em, src, mode={"vit": capi.MB_VITERBI, "cnt": 3, "tb": 4, "cntroll": 3 + 16, "fwdroll": 16}.get(mode, capi.MB_FORWARD), backward=(mode == "bwd"),
                      closure=(mode not in ("vit", "tb")), G=G)
full = src.replace(".hip", "_full.hip")
open(full, "w").write("#include <hip/hip_runtime.h>\n" + open(src).read())
asm = src.replace(".hip", ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only",
                       "-S", "-o", asm, full], stderr=None if os.environ.get("JIT_DUMP_VERBOSE") else subprocess.DEVNULL)
print(open(src).readline().strip())
for line in open(asm):
    if any(k in line for k in (".sgpr_count", ".vgpr_count", ".vgpr_spill_count", "; Occupancy", "; ScratchSize")):
        print(line.strip())
print(asm)

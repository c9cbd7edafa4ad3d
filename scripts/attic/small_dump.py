"""Dump the small-machine family's generated kernel for a preset and cross-compile it to gfx950 ISA (no GPU needed).

usage: python scripts/small_dump.py <preset> [mode sum|max|tb|cnt] [fwd|bwd] [mat|roll] [outdir]
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
preset = sys.argv[1]; mode = sys.argv[2] if len(sys.argv) > 2 else "sum"
direction = sys.argv[3] if len(sys.argv) > 3 else "fwd"; mat = (sys.argv[4] if len(sys.argv) > 4 else "mat") == "mat"
out = sys.argv[5] if len(sys.argv) > 5 else "/tmp/jit"
envelopes = os.environ.get("SMALL_DUMP_ENV", "0") == "1"
os.makedirs(out, exist_ok=True)
path = preset if os.path.exists(preset) else "tests/golden/preset/%s.json" % preset
m = Machine.fromFile(path); em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
src = os.path.join(out, "small_%s_%s_%s_%s.hip" % (os.path.basename(preset).replace(".json", ""), mode, direction, "mat" if mat else "roll"))
capi.debug_small_source(em, src, mode={"sum": 0, "max": 1, "tb": 2, "cnt": 3}[mode], backward=(direction == "bwd"), materialise=mat, envelopes=envelopes)
full = src.replace(".hip", "_full.hip")
open(full, "w").write("#include <hip/hip_runtime.h>\n" + open(src).read())
asm = src.replace(".hip", ".s")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only",
                    "-S", "-o", asm, full], stderr=subprocess.PIPE, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
print(open(src).readline().strip())
for line in open(asm):
    if any(k in line for k in (".sgpr_count", ".vgpr_count", ".vgpr_spill_count", "; Occupancy", "; ScratchSize")):
        print(line.strip())
print(asm)

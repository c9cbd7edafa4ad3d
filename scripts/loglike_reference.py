"""Per-pair Forward log-likelihoods of the bench's synthetic pairs (pair k: seed 1000 * 4 + k), computed on ONE GPU and committed as
profiles/<tag>_loglike_per_pair.json: what every rank of a multi-GPU bench run compares its shard's checksum with (bench.py,
extra.checks) -- the first contact with an 8-GPU node then says at once whether every rank computed its own pairs on its own device.

usage: python scripts/loglike_reference.py <out.json> [pairs=2048] [preset=psw2dna] [inlen=487] [outlen=10000]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
preset = sys.argv[3] if len(sys.argv) > 3 else "psw2dna"; il = int(sys.argv[4]) if len(sys.argv) > 4 else 487; ol = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
capi.set_device(0)
em = EvaluatedMachine.fromMachine(Machine.fromFile(os.path.join(ROOT, "tests", "golden", "preset", preset + ".json")), None, useDefaults=True)
dm = capi.DeviceMachine(em)
ll = []
for first in range(0, n, 256):
    b = capi.DeviceBatch(dm, *synth_batch(4, min(256, n - first), il, ol, em.nInTok, em.nOutTok, first=first))
    ll += [float(x) for x in b.forward(capi.MB_MATERIALISE)]      # the bench's default mode
    b.close()
json.dump({"preset": preset, "inlen": il, "outlen": ol, "mode": "materialise", "seed": "1000 * 4 + k", "pairs": n, "loglike": ll,
           "checksum_first_256": float(sum(ll[:256]))}, open(out, "w"))
print(out, n, sum(ll[:256]))

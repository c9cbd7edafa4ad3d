"""Does ncclCommInitRank survive a process that has already done heavy work through the library (cached workspaces, hiprtc
modules, a second HIP runtime from a late `import torch`)?  Prints the free device memory before the communicator is formed."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqgen import synth_batch
em = EvaluatedMachine.fromMachine(Machine.fromFile("tests/golden/preset/psw2dna.json"), None, useDefaults=True)
dm = capi.DeviceMachine(em)
b = capi.DeviceBatch(dm, *synth_batch(4, 96, 487, 10000, em.nInTok, em.nOutTok))
print("loglike", float(np.sum(b.forward(capi.MB_MATERIALISE))), flush=True)      # the matrices of the batch: ~100 GB of pool, kept cached
print("viterbi", float(np.sum(b.viterbi()[0])), flush=True)
if "torch" in sys.argv:
    import torch; print("torch.cuda.is_available()", torch.cuda.is_available(), flush=True)
hip = C.CDLL("libamdhip64.so")
free, tot = C.c_size_t(), C.c_size_t()
hip.hipMemGetInfo(C.byref(free), C.byref(tot)); print("free %.1f GB of %.1f" % (free.value / 1e9, tot.value / 1e9), flush=True)
c = capi.Comm(capi.Comm.unique_id(), 1, 0)
print("comm ok", c.allreduce_counts(np.arange(3.0), -1.0), flush=True)
c.close()

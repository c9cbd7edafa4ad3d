import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from machineboss_amd import capi
from randmachine import random_machine, random_seq
order = sys.argv[1]
def use_lib():
    em = random_machine(8, 2, 2, 1)
    dm = capi.DeviceMachine(em)
    rng = np.random.RandomState(0)
    b = capi.DeviceBatch.from_pairs(dm, [(random_seq(rng, 10, 2), random_seq(rng, 12, 2))])
    print("ll", b.forward(capi.MB_ROLLING))
def comm():
    c = capi.Comm(capi.Comm.unique_id(), 1, 0); print("comm ok"); c.close()
for ch in order:
    if ch == "l": use_lib()
    if ch == "t":
        import torch; print("torch", torch.__version__, torch.cuda.is_available())
    if ch == "d":
        import torch.distributed as dist; print("dist available", dist.is_available())
    if ch == "c":
        try: comm()
        except Exception as e: print("comm FAILED:", e)

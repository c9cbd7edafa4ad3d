"""BASELINE config 3: Forward-Backward EM (`boss --train`) on preset/protpsw.json, `pairs` x 400-aa synthetic pairs per GPU.
One EM iteration = weights -> device (mb_machine_set_weights), Backward + fused Forward/count sweep over the rank's
pairs, ONE all-reduce of nTransitions+1 doubles (RCCL when launched with torchrun), closed-form M-step on the host.
usage: python scripts/bench_train.py [pairs_per_gpu=1024] [len=400] [iterations=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
import torch
torch.cuda.set_device(local)
reduce = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from machineboss_amd.shard import allreduce_counts
    reduce = lambda c, ll: allreduce_counts(c, ll, "cuda")
from machineboss_amd import capi, fitter as F
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from machineboss_amd.seqpair import SeqPair
from machineboss_amd.seqgen import synth_tokens
capi.set_device(local)
m = Machine.fromFile("tests/golden/preset/protpsw.json")
em0 = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
data = []
for k in range(rank * pairs, (rank + 1) * pairs):
    x, y = synth_tokens(3000 + k, L, L, em0.nInTok, em0.nOutTok)
    data.append(SeqPair(em0.inputTokenizer.detokenize(x), em0.outputTokenizer.detokenize(y)))
F.MaxEMIterations = iters          # bounded run: time per iteration is what is measured
F.MinEMImprovement = -1e300
fit = F.MachineFitter(m)
t0 = time.perf_counter(); params = fit.fit(data, reduce=reduce); dt = time.perf_counter() - t0
n_it = len(fit.log)
cells = world * pairs * (L + 1) * (L + 1) * em0.nStates
if rank == 0:
    print("EM: %d iterations in %.2f s = %.1f ms/iteration; %.1f G lattice-cells/s (x2 matrices) over %d GPU(s); loglike %s"
          % (n_it, dt, dt / n_it * 1e3, cells * n_it / dt / 1e9, world, ["%.2f" % x for x in fit.log]))
if rank == 0 and os.environ.get("MB_TRAIN_PROFILE"):
    # breakdown of one steady-state iteration
    from machineboss_amd.dp import MachineCounts, _device_machine
    allp = dict(m.funcs); allp.update(params)
    t = time.perf_counter(); ev = EvaluatedMachine.fromMachine(m, allp); t_eval = time.perf_counter() - t
    dm = _device_machine(ev); batch = MachineCounts.deviceBatch(ev, data)
    mc = MachineCounts(ev); mc.addDeviceBatch(batch)
    t = time.perf_counter(); dm.set_weights(ev.logWeight); t_set = time.perf_counter() - t
    t = time.perf_counter(); mc = MachineCounts(ev); mc.addDeviceBatch(batch); t_e = time.perf_counter() - t
    t = time.perf_counter(); F.MachineObjective(m, mc, fit.constraints, {}).optimize(params); t_m = time.perf_counter() - t
    print("per iteration: eval weights %.1f ms, set_weights %.1f ms, E-step %.1f ms (device %.1f ms), M-step %.1f ms"
          % (t_eval * 1e3, t_set * 1e3, t_e * 1e3, capi.last_device_ms(), t_m * 1e3))

"""Per-transition accuracy of the one-tape E-step against the EXACT oracle: the 20-node profile composite of config 5
(5 063 states), sequences of several lengths and seeds, under --use-defaults parameters and under a NON-UNIFORM set (every norm group a
random point of its simplex, every prob random in (0.05, 0.95): no ties, another dynamic range).  Prints the largest relative deviation
(counts below 1e-3 taken as 1e-3) per case.

usage: python scripts/count_accuracy_onetape.py [L,seed,params ...]     e.g. 50000,2050,uniform 50000,7,random 12000,3,random"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi, algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
from oracle import oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda *p: os.path.join(ROOT, "tests", "golden", *p)


def profile_machine(nodes=20):
    P = lambda n: Machine.fromFile(G("preset", n + ".json"))
    h = HmmerModel.fromFile(G("hmmer", "fn3.hmm")).truncated(nodes)
    return A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])


def random_params(m, seed):
    """A non-uniform parameter set inside the machine's constraints (src/constraints.h): norm groups on their simplex, probs in (0.05, 0.95)."""
    rng = np.random.RandomState(seed)
    p = dict(m.cons.defaultParams())
    for group in m.cons.norm:
        w = rng.dirichlet(np.full(len(group), 2.0)) * 0.9 + 0.1 / len(group)
        for name, v in zip(group, w): p[name] = float(v)
    for name in m.cons.prob: p[name] = float(rng.uniform(0.05, 0.95))
    q = dict(m.funcs); q.update(p)
    return q


def main():
    cases = [c.split(",") for c in sys.argv[1:]] or [["50000", "2050", "uniform"], ["50000", "7", "random"], ["12000", "3", "random"]]
    oracle.build(); capi.set_device(0)
    m = profile_machine()
    ems = {"uniform": EvaluatedMachine.fromMachine(m, None, useDefaults=True)}
    for L, seed, par in cases:
        L, seed = int(L), int(seed)
        if par not in ems: ems[par] = EvaluatedMachine.fromMachine(m, random_params(m, 99))
        em = ems[par]
        om = oracle.OracleMachine(em); dm = capi.DeviceMachine(em)
        x = np.zeros(0, np.int32); y = np.random.RandomState(seed).randint(1, 4, size=L).astype(np.int32)
        b = capi.DeviceBatch.from_pairs(dm, [(x, y)])
        t0 = time.time(); counts, s, ll = b.counts(); t1 = time.time()
        ref = np.zeros(em.nTransitions); llo = om.counts_add(x, y, ref, oracle.SUM_EXACT)
        dev = np.abs(counts - ref) / np.maximum(np.abs(ref), 1e-3)
        inv = counts[np.asarray(em.outTok) != 0].sum() / L
        print("L=%d seed=%d params=%s: loglike rel err %.2e, largest per-transition count deviation %.3g (99.9 %% quantile %.3g), symbol-count invariant %.2e, E-step %.0f ms, kernel %s"
              % (L, seed, par, abs(ll[0] - llo) / abs(llo), dev.max(), np.quantile(dev, 0.999), abs(inv - 1), (t1 - t0) * 1e3, capi.last_kernel_name()), flush=True)
        b.close(); dm.close()


if __name__ == "__main__":
    main()

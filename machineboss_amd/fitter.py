"""Baum-Welch training: the callers either side of the count sweep (`boss --train`).

Mirrors /root/reference/src/fitter.{h,cpp} (EM loop) and the M-step of src/counts.cpp:117-295
(``MachineObjective``): minimise  E(theta) = - sum_e count[e] * log w_e(theta)  under the constraints
(``norm`` groups sum to one, ``prob`` parameters in [0,1], ``rate`` parameters >= 0).

The E-step -- all the arithmetic that scales with the data -- is ``MachineCounts`` on the GPU (dp.py); across
ranks it ends in ONE all-reduce of nTransitions + 1 doubles per iteration (shard.allreduce_counts, RCCL).
The M-step touches nTransitions numbers and stays on the host.  Where every transition weight is a product of
parameters, complements ``{"not": p}`` and constants -- true of every preset -- the maximiser is the closed form
``p = A / (A + B)`` / ``p_i = A_i / sum_j A_j`` (A, B = expected usage of p and of not-p); otherwise the
reference's own re-parameterisation (p_i = (1 - z_i) prod_{k<i} z_k, z = exp(-x^2); rate = x^2,
src/counts.cpp:139-170) is minimised with BFGS (scipy) -- the reference uses GSL's vector_bfgs2 on the same
function, and its tests pin the result to 4 significant digits.
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from .dp import MachineCounts, _deriv, _device_machine
from .evalmachine import EvaluatedMachine
from .machine import Constraints, Machine, MachineError, evalWeight, weightParams
from .seqpair import Envelope, SeqPair

MaxEMIterations = 1000      # src/fitter.cpp:6
MinEMImprovement = .001     # src/fitter.cpp:7


def combineConstraints(a: Constraints, b: Constraints) -> Constraints:
    """Constraints::combine (src/constraints.cpp)."""
    return Constraints(prob=list(a.prob) + list(b.prob), norm=[list(g) for g in a.norm] + [list(g) for g in b.norm],
                       rate=list(a.rate) + list(b.rate))


def _factors(w: Any, defs: Dict[str, Any], free: set, out: Dict[str, List[float]], sign: float = 1.0) -> bool:
    """Decompose log w into  sum_p (a_p log p + b_p log(1-p)) + const  when w is a product/quotient of free parameters,
    their complements and constants.  out[p] = [a_p, b_p].  Returns False if w has any other shape."""
    if w is None or isinstance(w, (bool, int, float)):
        return True
    if isinstance(w, str):
        if w in free:
            out.setdefault(w, [0.0, 0.0])[0] += sign
            return True
        if w in defs:
            v = defs[w]
            if isinstance(v, (int, float)):
                return True
            return _factors(v, {k: x for k, x in defs.items() if k != w}, free, out, sign)
        raise MachineError("Parameter %s not defined" % w)
    if not isinstance(w, dict) or not w:
        return False
    op, args = next(iter(w.items()))
    if op == "*":
        return _factors(args[0], defs, free, out, sign) and _factors(args[1], defs, free, out, sign)
    if op == "/":
        return _factors(args[0], defs, free, out, sign) and _factors(args[1], defs, free, out, -sign)
    if op == "not":
        if isinstance(args, str) and args in free:
            out.setdefault(args, [0.0, 0.0])[1] += sign
            return True
        return not (weightParams(args, defs) & free)     # complement of a constant is a constant
    return not (weightParams(w, defs) & free)            # any other expression is fine if it involves no free parameter


class MachineObjective:
    """src/counts.h MachineObjective: the M-step objective for one set of counts."""

    def __init__(self, machine: Machine, counts: MachineCounts, constraints: Constraints, constants: Dict[str, Any]):
        self.machine = machine
        self.constraints = combineConstraints(machine.cons, constraints)
        self.constantDefs = dict(machine.funcs); self.constantDefs.update(constants)
        self.terms = []     # (count, weight expression) per transition
        e = 0
        for ms in machine.state:
            for t in ms.trans:
                self.terms.append((float(counts._flat[e]), t.weight)); e += 1
        self.free = set(self.constraints.prob) | set(self.constraints.rate) | {p for g in self.constraints.norm for p in g}

    # E(params) = - sum count * log w   (src/counts.cpp:122-131)
    def value(self, params: Dict[str, float]) -> float:
        defs = dict(self.constantDefs); defs.update(params)
        f = 0.0
        for c, w in self.terms:
            if c != 0.0:
                v = evalWeight(w, defs)
                f -= c * (math.log(v) if v > 0 else -math.inf)
        return f

    def _closed_form(self, seed: Dict[str, Any]) -> Optional[Dict[str, float]]:
        if self.constraints.rate:
            return None
        ab: Dict[str, List[float]] = {}
        for c, w in self.terms:
            local: Dict[str, List[float]] = {}
            if not _factors(w, self.constantDefs, self.free, local):
                return None
            for p, (a, b) in local.items():
                acc = ab.setdefault(p, [0.0, 0.0]); acc[0] += c * a; acc[1] += c * b
        out: Dict[str, float] = {}
        for p in self.constraints.prob:
            a, b = ab.get(p, [0.0, 0.0])
            if a < 0 or b < 0:
                return None
            out[p] = a / (a + b) if a + b > 0 else float(seed[p])
        for g in self.constraints.norm:
            if any(ab.get(p, [0.0, 0.0])[1] != 0.0 or ab.get(p, [0.0, 0.0])[0] < 0 for p in g):
                return None     # a complement of a normalised parameter: no closed form
            tot = sum(ab.get(p, [0.0, 0.0])[0] for p in g)
            for p in g:
                out[p] = ab.get(p, [0.0, 0.0])[0] / tot if tot > 0 else float(seed[p])
        return out

    # ---- the reference's transformed parameterisation (src/counts.cpp:139-170, 236-262) ---------------------------
    def _layout(self):
        idx: Dict[str, int] = {}
        for g in self.constraints.norm:
            for p in g[:-1]:
                idx[p] = len(idx)
        for p in self.constraints.prob:
            idx[p] = len(idx)
        for p in self.constraints.rate:
            idx[p] = len(idx)
        return idx

    def _to_params(self, x: np.ndarray, idx: Dict[str, int]) -> Dict[str, float]:
        out: Dict[str, float] = {}
        for g in self.constraints.norm:
            notPrev = 1.0
            for n, p in enumerate(g):
                if n + 1 == len(g):
                    out[p] = notPrev
                else:
                    z = math.exp(-x[idx[p]] ** 2)
                    out[p] = notPrev * (1.0 - z); notPrev *= z
        for p in self.constraints.prob:
            out[p] = math.exp(-x[idx[p]] ** 2)
        for p in self.constraints.rate:
            out[p] = x[idx[p]] ** 2
        return out

    def _grad_x(self, x: np.ndarray, idx: Dict[str, int], params: Dict[str, float], dE: Dict[str, float]) -> np.ndarray:
        """dE/dx from dE/dp through the transform of _to_params, analytically: with z_k = exp(-x_k^2) a normalised group has
        p_k = N_k (1 - z_k), N_(k+1) = N_k z_k, the last p = N, so dp_k/dz_k = -N_k and dp_q/dz_k = p_q / z_k for the later q."""
        g = np.zeros(len(idx))
        for grp in self.constraints.norm:
            tail = 0.0      # sum over the later parameters q of dE/dp_q p_q
            zs = [math.exp(-x[idx[p]] ** 2) for p in grp[:-1]]
            Ns = [1.0]
            for z in zs: Ns.append(Ns[-1] * z)
            tail = dE.get(grp[-1], 0.0) * params[grp[-1]]
            for k in range(len(grp) - 2, -1, -1):
                p = grp[k]; z = zs[k]
                dEdz = -Ns[k] * dE.get(p, 0.0) + (tail / z if z > 0.0 else 0.0)
                g[idx[p]] = dEdz * (-2.0 * x[idx[p]] * z)
                tail += dE.get(p, 0.0) * params[p]
        for p in self.constraints.prob:
            g[idx[p]] = dE.get(p, 0.0) * (-2.0 * x[idx[p]] * params[p])
        for p in self.constraints.rate:
            g[idx[p]] = dE.get(p, 0.0) * 2.0 * x[idx[p]]
        return g

    def _seed_x(self, seed: Dict[str, Any], idx: Dict[str, int]) -> np.ndarray:
        x = np.zeros(len(idx))
        for g in self.constraints.norm:
            pSum = 0.0
            for p in g[:-1]:
                v = float(seed[p]); z = 1 - v / (1 - pSum)
                x[idx[p]] = math.sqrt(-math.log(z)); pSum += v
        for p in self.constraints.prob:
            x[idx[p]] = math.sqrt(-math.log(float(seed[p])))
        for p in self.constraints.rate:
            x[idx[p]] = math.sqrt(float(seed[p]))
        return x

    def optimize(self, seed: Dict[str, Any]) -> Dict[str, Any]:
        """MachineObjective::optimize: returns ``seed`` with the constrained parameters replaced by the maximiser."""
        final = dict(seed)
        cf = self._closed_form(seed)
        if cf is not None:
            final.update(cf)
            return final
        from scipy.optimize import minimize
        idx = self._layout()
        if not idx:
            return final

        # The objective and its gradient through ONE program compiled for the machine (evalmachine.CompiledWeights: values forward, adjoints
        # backward, level by level in numpy) instead of one symbolic evaluation per transition and parameter: the 5 063-state composition
        # of BASELINE config 5 (14 691 terms, 84 parameters, 1 905 terms that are sums) takes 0.18 s per VALUE the scalar way.
        # MB_FITTER_COMPILED=0: the scalar way (the parity test of the two).
        import os as _os
        if _os.environ.get("MB_FITTER_COMPILED", "1") != "0":
            from .evalmachine import CompiledWeights
            # (compiled once per machine, free-parameter set and constant definitions: every EM iteration makes a new MachineObjective)
            key = (hash(tuple(id(t.weight) for ms in self.machine.state for t in ms.trans)), tuple(sorted(self.free)),
                   tuple(sorted((k, id(v) if isinstance(v, (dict, list)) else v) for k, v in self.constantDefs.items())))
            cached = getattr(self.machine, "_compiledObjective", None)
            if cached is None or cached[0] != key:
                cached = (key, CompiledWeights(self.machine, expand=self.constantDefs, keep=self.free))
                self.machine._compiledObjective = cached
            cw = cached[1]
            cvec = np.array([c for c, _ in self.terms], np.float64)

            def both(x):
                params = self._to_params(x, idx)
                defs = dict(self.constantDefs); defs.update(params)
                try:
                    E, dE = cw.objective(cvec, defs)
                except (ValueError, ZeroDivisionError, MachineError):
                    return 1e300, np.zeros(len(idx))
                if dE is None or not math.isfinite(E):
                    return 1e300, np.zeros(len(idx))
                return E, self._grad_x(x, idx, params, dE)
            res = minimize(both, self._seed_x(seed, idx), jac=True, method="BFGS", options={"gtol": 1e-7, "maxiter": 1000})
            final.update(self._to_params(res.x, idx))
            return final

        def f(x):
            try:
                v = self.value(self._to_params(x, idx))
            except (ValueError, ZeroDivisionError):
                return 1e300
            return v if math.isfinite(v) else 1e300

        def grad(x):
            params = self._to_params(x, idx)
            defs = dict(self.constantDefs); defs.update(params)
            dE = {p: 0.0 for p in params}
            for c, w in self.terms:
                if c == 0.0:
                    continue
                v = evalWeight(w, defs)
                for p in weightParams(w, self.constantDefs) & set(params):
                    dE[p] -= c * _deriv(w, defs, p) / v
            g = np.zeros(len(idx))
            eps = 1e-7
            for p, j in idx.items():            # dp/dx_j by central differences of the (cheap, exact) transform
                xp = x.copy(); xm = x.copy(); xp[j] += eps; xm[j] -= eps
                pp, pm = self._to_params(xp, idx), self._to_params(xm, idx)
                g[j] = sum(dE[q] * (pp[q] - pm[q]) / (2 * eps) for q in params)
            return g
        res = minimize(f, self._seed_x(seed, idx), jac=grad, method="BFGS", options={"gtol": 1e-7, "maxiter": 1000})
        final.update(self._to_params(res.x, idx))
        return final


class MachineFitter:
    """src/fitter.h: Baum-Welch over a SeqPairList."""

    def __init__(self, machine: Machine, constraints: Optional[Constraints] = None, constants: Optional[Dict[str, Any]] = None,
                 seed: Optional[Dict[str, Any]] = None):
        self.machine = machine
        self.constraints = constraints or Constraints()
        self.constants = dict(constants or {})
        self.seed = dict(seed) if seed is not None else self.allConstraints().defaultParams()
        self.log: List[float] = []       # log-likelihood per iteration ("Baum-Welch iteration #k", src/fitter.cpp:31)
        self.timing: List[Dict[str, float]] = []      # per iteration, milliseconds: weight evaluation, set_weights, E-step (wall / device), M-step

    def allConstraints(self) -> Constraints:
        return combineConstraints(self.machine.cons, self.constraints)

    def fit(self, trainingSet: Sequence[SeqPair], width: Optional[int] = None,
            reduce: Optional[Callable[[np.ndarray, float], Any]] = None) -> Dict[str, Any]:
        """MachineFitter::fit (src/fitter.cpp:15-49).  ``width`` is `--wiggle-room`: accepted and, like in the reference,
        without effect (quirk Q1: the matrices always use Envelope(seqPair)).  ``reduce(counts, loglike)`` sums the
        E-step statistics over ranks when the training set is sharded (shard.allreduce_counts); None = single process."""
        envelopes = [Envelope(sp) if width is None else Envelope(sp, width) for sp in trainingSet]
        if len(envelopes) != len(trainingSet):
            raise MachineError("Envelope/training set mismatch")
        params = dict(self.seed)
        prev = 0.0
        self.log = []
        it = 0
        dm = None
        batch = None
        import time
        from . import capi as _capi
        self.timing = []
        while True:
            tm = {}
            t0 = time.perf_counter()
            allParams = dict(self.machine.funcs); allParams.update(self.constants); allParams.update(params)
            # (later iterations: the same topology under new parameters -- the weight expressions through a program compiled once,
            #  evalmachine.CompiledWeights; the reference rebuilds the whole EvaluatedMachine, src/fitter.cpp:28-29)
            ev = EvaluatedMachine.fromMachine(self.machine, allParams) if dm is None else ev.reweighted(self.machine, allParams)
            tm["eval_ms"] = (time.perf_counter() - t0) * 1e3; t0 = time.perf_counter()
            # the topology is uploaded (and its kernels specialised) once; later iterations only send new log-weights
            # (mb_machine_set_weights) -- the reference rebuilds its EvaluatedMachine every iteration (src/fitter.cpp:28-29)
            if dm is None:
                dm = _device_machine(ev)
                batch = MachineCounts.deviceBatch(ev, trainingSet)      # tokenised once, resident in HBM for all iterations
            else:
                dm.set_weights(ev.logWeight)
                ev._device = dm
            tm["set_weights_ms"] = (time.perf_counter() - t0) * 1e3; t0 = time.perf_counter()      # (first iteration: upload, kernel specialisation, tokenisation)
            counts = MachineCounts(ev)
            counts.addDeviceBatch(batch)
            tm["estep_ms"] = (time.perf_counter() - t0) * 1e3; tm["estep_device_ms"] = _capi.last_device_ms()
            self.timing.append(tm)
            if reduce is not None:
                _, ll = reduce(counts._flat, counts.loglike)
                counts.loglike = ll
            self.log.append(counts.loglike)
            if it > 0:
                if it == MaxEMIterations:
                    break
                improvement = (counts.loglike - prev) / abs(prev)
                if improvement < MinEMImprovement:
                    break
            t0 = time.perf_counter()
            params = MachineObjective(self.machine, counts, self.constraints, self.constants).optimize(params)
            tm["mstep_ms"] = (time.perf_counter() - t0) * 1e3
            prev = counts.loglike
            it += 1
        return params

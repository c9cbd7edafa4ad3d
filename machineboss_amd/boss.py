"""`boss`-compatible command line for the inference surface, backed by the HIP engine.

    python -m machineboss_amd.boss MACHINE.json [--preset NAME] [-P params.json] [-F funcs.json] [-N constraints.json]
           [-D seqpairs.json] [--input-chars S] [--output-chars S] [--input-fasta F] [--output-fasta F]
           [--input-json F] [--output-json F] [--use-defaults] [-L] [-V] [-A] [-C] [-T] [-R width]

Restates the data-handling and inference section of /root/reference/target/boss.cpp:716-847 -- how sequences are
collected into pairs, how parameters are assembled, and the exact output text of --loglike / --viterbi / --align /
--counts / --train -- around the batched GPU calls.  The real `boss` needs Boost and cannot run on the GPU box; the
machine-expression language of its command line is reduced to what assembles the benchmark machines: several
transducer files / presets on one command line are COMPOSED (algebra.py = Machine::compose); the other operators
(concatenate, union, Kleene closures, ...) are not here.

Numbers print like C++ `ostream << double` (6 significant digits, target/boss.cpp:794-807 via src/jsonio.h:14-22),
parameters with 15 (src/weight.cpp:483).  Work is batched: all pairs go through one device call per mode; with
torch.distributed initialised (torchrun) the pairs are sharded over ranks, --train/--counts all-reduce their counts
(RCCL) and rank 0 prints.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
from typing import Any, Dict, List, Optional, Tuple

from .evalmachine import EvaluatedMachine
from .machine import Constraints, Machine, MachineError
from .seqpair import SeqPair, seqPairListFromJson

PRESET_DIRS = [os.environ.get("MB_PRESET_DIR", ""), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                  "tests", "golden", "preset")]


def fmt(x: float) -> str:
    """toInfinitySafeString (src/jsonio.h:14-22): `out << x` = %g with 6 significant digits."""
    if x == math.inf:
        return '"Infinity"'
    if x == -math.inf:
        return '"-Infinity"'
    return "%g" % x


def fmtParam(x: Any) -> str:
    """WeightAlgebra::toJsonStream for constants (src/weight.cpp:471-484): 0, 1, ints, doubles at 15 digits."""
    if isinstance(x, bool):
        return "1" if x else "0"
    if isinstance(x, (int, float)):
        if x == 0:
            return "0"
        if x == 1:
            return "1"
        if isinstance(x, int):
            return str(x)
        return "%.15g" % x
    return json.dumps(x, separators=(",", ":"))


def escaped(s: str) -> str:
    return json.dumps(s)[1:-1]


def readFasta(path: str) -> List[Tuple[str, str]]:
    out: List[Tuple[str, str]] = []
    name, seq = None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(seq)))
                name, seq = line[1:].split()[0] if len(line) > 1 else "", []
            elif line and name is not None:
                seq.append(line)
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def seqPairJson(sp: SeqPair) -> str:
    """SeqPair::writeJson (src/seqpair.cpp:40-58)."""
    named = lambda n, s: '{"name":"%s","sequence":[%s]}' % (n, ",".join('"%s"' % x for x in s))
    out = '{"input":' + named(sp.inputName, sp.input) + ',"output":' + named(sp.outputName, sp.output)
    if sp.alignment:
        out += ',"alignment":[' + ",".join('["%s","%s"]' % (escaped(a), escaped(b)) for a, b in sp.alignment) + "]"
    if sp.metadata is not None:
        out += ',"meta":' + json.dumps(sp.metadata, separators=(",", ":"), sort_keys=True)
    return out + "}"


def pathJson(m: Machine, path) -> dict:
    """MachinePath::writeJson (src/machine.cpp:1982-2000) as the JSON object stored under meta.path."""
    j: Dict[str, Any] = {"start": m.startState()}
    if m.state[m.startState()].name is not None:
        j["id"] = m.state[m.startState()].name
    trans = []
    for t in path.trans:
        tj: Dict[str, Any] = {"to": t.dest}
        if m.state[t.dest].name is not None:
            tj["id"] = m.state[t.dest].name
        if t.inp:
            tj["in"] = t.inp
        if t.out:
            tj["out"] = t.out
        trans.append(tj)
    j["trans"] = trans
    return j


def seqPairFromPath(m: Machine, path, inputName: str, outputName: str) -> SeqPair:
    """SeqPair::seqPairFromPath (src/seqpair.cpp:60-89)."""
    ali = [(t.inp, t.out) for t in path.trans if t.inp or t.out]
    return SeqPair([a for a, _ in ali if a], [b for _, b in ali if b], inputName, outputName, ali, {"path": pathJson(m, path)})


def buildParser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="boss", description=__doc__.split("\n\n")[0])
    ap.add_argument("machine", nargs="*", help="transducer JSON file")
    ap.add_argument("--preset", action="append", default=[], help="preset name (dnapsw, protpsw, psw2dna, translate); repeatable")
    ap.add_argument("-H", "--hmmer", help="generator from a HMMER3 model file, local alignment mode (target/boss.cpp:574-579); leftmost")
    ap.add_argument("--hmmer-global", help="the same in global alignment mode")
    ap.add_argument("--hmmer-plan7", help="Plan7 generator (single hit, N/C flanks)")
    ap.add_argument("--hmmer-multihit", help="Plan7 generator with the J loop")
    ap.add_argument("--generate-chars", help="compose a generator of this sequence in front of the machine(s)")
    ap.add_argument("--recognize-chars", help="compose a recogniser of this sequence behind the machine(s)")
    ap.add_argument("-P", "--params", action="append", default=[])
    ap.add_argument("-F", "--functions", action="append", default=[])
    ap.add_argument("-N", "--constraints", action="append", default=[])
    ap.add_argument("-D", "--data", action="append", default=[])
    ap.add_argument("--use-defaults", action="store_true")
    ap.add_argument("--input-chars"); ap.add_argument("--output-chars")
    ap.add_argument("--input-fasta"); ap.add_argument("--output-fasta")
    ap.add_argument("--input-json"); ap.add_argument("--output-json")
    ap.add_argument("-L", "--loglike", action="store_true")
    ap.add_argument("-V", "--viterbi", action="store_true")
    ap.add_argument("-A", "--align", action="store_true")
    ap.add_argument("-C", "--counts", action="store_true")
    ap.add_argument("-T", "--train", action="store_true")
    ap.add_argument("-R", "--wiggle-room", type=int)
    return ap


def loadPreset(name: str) -> Machine:
    for d in PRESET_DIRS:
        p = os.path.join(d, name + ".json")
        if d and os.path.exists(p):
            return Machine.fromFile(p)
    raise MachineError("Unknown preset %s" % name)


def loadMachine(args) -> Machine:
    """Several machines on one command line are composed, right to left (target/boss.cpp:268-276, 628-634); presets come
    first, in the order given."""
    from .algebra import composeAll, generator, recognizer
    machines = [loadPreset(n) for n in args.preset] + [Machine.fromFile(f) for f in args.machine]
    from .hmmer import HmmerModel               # profile generators go in front of the transducers (target/boss.cpp:574-600)
    for path, build in ((args.hmmer, lambda h: h.machine(True)), (args.hmmer_global, lambda h: h.machine(False)),
                        (args.hmmer_plan7, lambda h: h.plan7Machine(False)), (args.hmmer_multihit, lambda h: h.plan7Machine(True))):
        if path is not None:
            machines.insert(0, build(HmmerModel.fromFile(path)))
    if args.generate_chars is not None:       # leftmost: a generator of the sequence (target/boss.cpp:362-364)
        machines.insert(0, generator(list(args.generate_chars), args.generate_chars))
    if args.recognize_chars is not None:      # rightmost: a recogniser of the sequence (target/boss.cpp:384-386)
        machines.append(recognizer(list(args.recognize_chars), args.recognize_chars))
    if not machines:
        raise MachineError("Please specify a transducer")
    return composeAll(machines)


def collectData(args, machine: Machine, inferenceRequested: bool) -> List[SeqPair]:
    """target/boss.cpp:716-773."""
    data: List[SeqPair] = []
    for f in args.data:
        data += seqPairListFromJson(json.load(open(f)))
    inSeqs: List[Tuple[str, List[str]]] = []
    outSeqs: List[Tuple[str, List[str]]] = []
    if args.input_fasta:
        inSeqs += [(n, list(s)) for n, s in readFasta(args.input_fasta)]
    if args.input_chars is not None:
        inSeqs.append((args.input_chars, list(args.input_chars)))
    if args.output_fasta:
        outSeqs += [(n, list(s)) for n, s in readFasta(args.output_fasta)]
    if args.output_chars is not None:
        outSeqs.append((args.output_chars, list(args.output_chars)))
    if args.input_json:
        j = json.load(open(args.input_json)); inSeqs.append((j.get("name", ""), list(j["sequence"])))
    if args.output_json:
        j = json.load(open(args.output_json)); outSeqs.append((j.get("name", ""), list(j["sequence"])))
    inputEmpty, outputEmpty = not machine.inputAlphabet(), not machine.outputAlphabet()
    if not inSeqs and inputEmpty and ((outputEmpty and inferenceRequested) or outSeqs):
        inSeqs.append(("", []))
    if not outSeqs and inSeqs and outputEmpty:
        outSeqs.append(("", []))
    for iname, iseq in inSeqs:
        for oname, oseq in outSeqs:
            data.append(SeqPair(iseq, oseq, iname, oname))
    if inferenceRequested and not data and inputEmpty and outputEmpty:
        data.append(SeqPair([], [], "", ""))
    return data


_GROUP = None      # shard.RankGroup of this process when launched as one rank of several (main() opens it)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _reduce_counts(counts, loglike):
    """Sum of the E-step statistics over the ranks (MachineCounts::operator+=, src/counts.cpp:66-71): RCCL through the C-ABI when
    main() opened the ranks (shard.RankGroup), torch.distributed's own group when a host program opened one."""
    if _GROUP is not None:
        return _GROUP.allreduce_counts(counts, loglike)
    from .shard import allreduce_counts
    import torch
    return allreduce_counts(counts, loglike, "cuda" if (torch.cuda.is_available() and _dist().get_backend() == "nccl") else "cpu")


def _shard(data: List[Any], machine, world: int) -> List[List[int]]:
    """Which pairs each rank takes: sorted by DP cell count and dealt greedily to the least loaded rank
    (shard.lpt_assign) -- pairs are independent units (the `for seqPair` loops of target/boss.cpp:796,826), and a ragged list
    dealt round-robin leaves the rank that drew the long pairs working alone.  Deterministic: every rank computes the same."""
    from .shard import lpt_assign
    nStates = len(machine.state)
    cells = [(len(sp.input) + 1) * (len(sp.output) + 1) * nStates for sp in data]
    return lpt_assign(cells, world)


def _gather_in_order(local: List[Any], n: int, rank: int, world: int, owned: Optional[List[List[int]]] = None) -> List[Any]:
    """Results of the sharded pairs back in input order on every rank (no data-path collective: host gather).
    owned[r] = indices of the pairs rank r took, in the order it processed them."""
    dist = _dist()
    if dist is None or world == 1:
        return local
    parts: List[Any] = [None] * world
    dist.all_gather_object(parts, local)
    out: List[Any] = [None] * n
    for r in range(world):
        for k, v in zip(owned[r] if owned is not None else range(r, n, world), parts[r]):
            out[k] = v
    return out


def run(argv: Optional[List[str]] = None, out=None) -> int:
    out = out or sys.stdout
    args = buildParser().parse_args(argv)
    machine = loadMachine(args)
    inference = args.loglike or args.viterbi or args.align or args.counts or args.train
    data = collectData(args, machine, inference)
    gotData = bool(data)
    noIO = not machine.inputAlphabet() and not machine.outputAlphabet()
    if gotData and not inference:
        raise MachineError("No point in specifying input/output data without --train, --loglike, --counts, --align")
    funcs: Dict[str, Any] = {}
    for f in args.functions:
        funcs.update(json.load(open(f)))
    seed: Dict[str, Any] = {}
    for f in args.params:
        seed.update(json.load(open(f)))
    constraints = Constraints()
    for f in args.constraints:
        c = Constraints.fromJson(json.load(open(f)))
        constraints = Constraints(constraints.prob + c.prob, constraints.norm + c.norm, constraints.rate + c.rate)

    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    owned = _shard(data, machine, world) if world > 1 else [list(range(len(data)))]
    mine = [data[k] for k in owned[rank]]
    emit = (lambda s: out.write(s)) if rank == 0 else (lambda s: None)

    from . import dp
    if args.train:
        from .fitter import MachineFitter, combineConstraints
        if not ((args.constraints or not machine.cons.empty()) and (gotData or noIO)):
            raise MachineError("To fit parameters, please specify a constraints file and (for machines with input/output) a data file")
        fitter = MachineFitter(machine, constraints, funcs)
        sd = combineConstraints(machine.cons, constraints).defaultParams(); sd.update(seed)
        fitter.seed = sd
        reduce = _reduce_counts if world > 1 else None
        params = fitter.fit(mine, args.wiggle_room, reduce)
        emit("{" + ",".join('"%s":%s' % (escaped(k), fmtParam(params[k])) for k in sorted(params)) + "}\n")
    else:
        params = dict(funcs); params.update(seed)
        for k, v in machine.getParamDefs(args.use_defaults).items():
            params.setdefault(k, v)

    if args.loglike:
        ev = EvaluatedMachine.fromMachine(machine, params)
        ll = _gather_in_order(dp.forwardLogLikeBatch(ev, mine, rolling=True), len(data), rank, world, owned)
        emit("[" + ",\n ".join('["%s","%s",%s]' % (escaped(sp.inputName), escaped(sp.outputName), fmt(x))
                                for sp, x in zip(data, ll)) + "]\n")

    if args.counts:
        ev = EvaluatedMachine.fromMachine(machine, params)
        counts = dp.MachineCounts(ev, mine)
        if world > 1:
            _, counts.loglike = _reduce_counts(counts._flat, counts.loglike)
        pc = counts.paramCounts(machine, params)
        emit("{" + ",".join('"%s":%s' % (escaped(k), "%g" % pc[k]) for k in sorted(pc)) + "}\n")

    if args.align or args.viterbi:
        if not gotData:
            raise MachineError("To align sequences, please specify a data file")
        ev = EvaluatedMachine.fromMachine(machine, params)
        res = dp.viterbiBatch(ev, machine, mine)
        res = _gather_in_order(res, len(data), rank, world, owned)
        if args.viterbi:
            emit("[" + ",\n ".join('["%s","%s",%s]' % (escaped(sp.inputName), escaped(sp.outputName), fmt(v))
                                    for sp, (v, _) in zip(data, res)) + "]\n")
        if args.align:
            aligned = [seqPairFromPath(machine, p, sp.inputName, sp.outputName) for sp, (v, p) in zip(data, res) if p is not None]
            emit("[" + ",\n ".join(seqPairJson(sp) for sp in aligned) + "]\n")
    return 0


def _init_ranks():
    """Launched under torchrun (WORLD_SIZE > 1): bind this rank to its GPU and open the ranks BEFORE the first GPU call, so that
    run() shards the pair list and reduces the counts (shard.RankGroup: rendezvous over gloo, the count reduction over RCCL
    through the C-ABI; MB_DIST_BACKEND = gloo puts several ranks on ONE GPU for the tests).  Returns the group if this call
    opened it (the caller closes it), else None."""
    global _GROUP
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or _dist() is not None:
        return None
    from .shard import RankGroup
    _GROUP = RankGroup.from_env()
    return _GROUP


def main(argv: Optional[List[str]] = None) -> int:
    opened = None
    try:
        opened = _init_ranks()
        return run(argv)
    except (MachineError, OSError, KeyError, ValueError) as e:   # main() of the reference prints what() and fails (boss.cpp:923-926)
        sys.stderr.write(str(e) + "\n")
        return 1
    finally:
        if opened is not None:
            global _GROUP
            opened.close(); _GROUP = None


if __name__ == "__main__":
    sys.exit(main())

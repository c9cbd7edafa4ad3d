"""Is there a dense block worth MFMA in BASELINE config 5 (HMMER profile . simple_introns . translate . dnapsw)?

The only dense contraction in the un-composed recurrence is profile emission x translation: for profile node k and a DNA
codon c,  score[k][c] = log sum_a  e_k(a) * P(c | a)   with e = the profile's K x 20 match (and insert) emission rows and
P = the 20 x 64 amino-acid -> codon table of `translate` (x dnapsw's substitution model).  That product does not depend on
the sequence: it is a K x 20 by 20 x 64 matrix product done ONCE per machine (here, on the host, by the composition), after
which a column of the DP reads score[k][observed codon] -- a table look-up.  Everything else in a column is the sparse scan
over profile nodes x intron / codon-phase states that the composed machine already is.  This script puts numbers on it.
usage: python scripts/plan7_dense_block.py [nodes=86]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import algebra as A
from machineboss_amd.hmmer import HmmerModel
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 86
P = lambda n: Machine.fromFile(os.path.join("tests", "golden", "preset", n + ".json"))
h = HmmerModel.fromFile(os.path.join("tests", "golden", "hmmer", "fn3.hmm")).truncated(nodes)
prof = EvaluatedMachine.fromMachine(h.machine(True), None, useDefaults=True)
tr = EvaluatedMachine.fromMachine(P("translate"), None, useDefaults=True)
m = A.composeLeftToRight([h.machine(True), P("simple_introns"), P("translate"), P("dnapsw")])
em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
K, nAA, nCodon = nodes, 20, 64
emit = int(np.sum(np.asarray(prof.outTok) != 0))
dense_flops_once = 2 * (2 * K) * nAA * nCodon          # match + insert rows of every node against the codon table
per_col_edges = em.nTransitions
sil = int(np.sum((np.asarray(em.inTok) == 0) & (np.asarray(em.outTok) == 0)))
print("profile: %d nodes, %d states, %d emitting transitions (= %d x 20 emission entries)" % (K, prof.nStates, emit, emit // 20))
print("translate: %d states, %d transitions; codon table 20 x 64" % (tr.nStates, tr.nTransitions))
print("composed machine: %d states, %d transitions (%d output-only, %d silent), %d silent levels" %
      (em.nStates, em.nTransitions, em.nTransitions - sil, sil, int(em.silentLevels().max()) + 1))
print("dense block: (2 x %d) x 20 by 20 x 64 = %d flops, ONCE per machine; per column it is a look-up of %d entries" % (K, dense_flops_once, 2 * K))
print("sparse scan: %d candidates per column (add + max / log-sum-exp each), edge density %.1e" % (per_col_edges, per_col_edges / float(em.nStates) ** 2))
print("if the dense product were redone per column on MFMA it would be %d flops against ~%d for the scan: %.1f %% of the work" %
      (dense_flops_once, 10 * per_col_edges, 100.0 * dense_flops_once / (10.0 * per_col_edges + dense_flops_once)))

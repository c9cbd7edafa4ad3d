// TEST SCAFFOLDING for the mock build only: the free functions of the reference's src/api.h:20-34.  In the reference tree
// src/api.cpp stays as it is and compiles against the replaced classes; the mock has no api.cpp, so the same six
// three-line wrappers are spelled here over the glue's typedefs for tests/cxx/test_glue.cpp to call.
#pragma once
#include "hipdp.h"

namespace MachineBoss {
inline double forwardLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  return ForwardMatrix(eval, seqPair).logLike();
}
inline double forwardLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair, const Envelope &env) {
  const EvaluatedMachine eval(machine, params);
  return ForwardMatrix(eval, seqPair, env).logLike();
}
inline double viterbiLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  return ViterbiMatrix(eval, seqPair).logLike();
}
inline MachinePath viterbiAlign(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  return ViterbiMatrix(eval, seqPair).path(machine);
}
inline MachineCounts forwardBackwardCounts(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  return MachineCounts(eval, seqPair);
}
inline MachineCounts forwardBackwardCounts(const Machine &machine, const Params &params, const SeqPairList &seqPairList) {
  const EvaluatedMachine eval(machine, params);
  return MachineCounts(eval, seqPairList);
}
}  // namespace MachineBoss

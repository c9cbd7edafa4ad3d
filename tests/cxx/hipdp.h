// hipdp.h -- the ONE glue header a Machine Boss maintainer adds: it binds the shim's class
// templates to the reference's own types, so that src/{forward,backward,viterbi,counts}.h can be replaced by
// `#include "hipdp.h"` and every caller (target/boss.cpp, src/api.cpp, src/fitter.cpp, src/machine.cpp's downsample,
// t/src/test*.cpp) compiles unchanged.  Compiled here against tests/cxx/mock_reference.h, in the reference tree against
// eval.h / seqpair.h / machine.h.
#pragma once
#include "mb_dp.hpp"

namespace MachineBossHIP {
template <> struct PathOf<MachineBoss::Machine> { typedef MachineBoss::MachinePath type; };   // paths come back as the reference's MachinePath
}

namespace MachineBoss {

typedef MachineBossHIP::ForwardMatrixT<EvaluatedMachine, SeqPair> ForwardMatrix;                         // src/forward.h:19-27
typedef MachineBossHIP::RollingOutputForwardMatrixT<EvaluatedMachine, SeqPair> RollingOutputForwardMatrix;   // src/forward.h:29
typedef MachineBossHIP::BackwardMatrixT<EvaluatedMachine, SeqPair> BackwardMatrix;                       // src/backward.h:10-59
typedef MachineBossHIP::ViterbiMatrixT<EvaluatedMachine, SeqPair> ViterbiMatrix;                         // src/viterbi.h:9-18
typedef MachineBossHIP::MachineCountsT<EvaluatedMachine, SeqPair, SeqPairList> MachineCounts;           // src/counts.h:11-25
template <class IndexMapper> using DPMatrix = MachineBossHIP::DPMatrixCore;                               // DPMatrix<IdentityIndexMapper>::TraceTerminator etc.
struct IdentityIndexMapper {};

// src/api.h:20-34 / src/api.cpp:32-58, verbatim bodies over the replaced classes
inline double forwardLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  const ForwardMatrix fwd(eval, seqPair);
  return fwd.logLike();
}
inline double forwardLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair, const Envelope &env) {
  const EvaluatedMachine eval(machine, params);
  const ForwardMatrix fwd(eval, seqPair, env);
  return fwd.logLike();
}
inline double viterbiLogLike(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  const ViterbiMatrix vit(eval, seqPair);
  return vit.logLike();
}
inline MachinePath viterbiAlign(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  const ViterbiMatrix vit(eval, seqPair);
  return vit.path(machine);
}
inline MachineCounts forwardBackwardCounts(const Machine &machine, const Params &params, const SeqPair &seqPair) {
  const EvaluatedMachine eval(machine, params);
  MachineCounts counts(eval, seqPair);
  return counts;
}
inline MachineCounts forwardBackwardCounts(const Machine &machine, const Params &params, const SeqPairList &seqPairList) {
  const EvaluatedMachine eval(machine, params);
  MachineCounts counts(eval, seqPairList);
  return counts;
}

}  // namespace MachineBoss

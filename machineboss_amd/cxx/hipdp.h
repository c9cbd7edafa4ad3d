// hipdp.h -- the ONE glue header a Machine Boss maintainer adds to src/ (INTEGRATION.md section 2): it binds the shim's
// class templates (mb_dp.hpp) to the reference's own types, so that src/{dpmatrix,forward,backward,viterbi}.h become
// `#include "hipdp.h"`, src/counts.h keeps MachineObjective and takes its MachineCounts from here, and every caller
// (target/boss.cpp, src/api.cpp, src/fitter.cpp, src/counts.cpp:117-295, src/machine.cpp's downsample, t/src/test*.cpp)
// compiles unchanged.  `apply_glue.py <machineboss tree>` writes those five headers; tests/test_cxx_glue.py runs it on
// /root/reference (when present) and compiles the reference's own t/src/test{forward,backward,counts,maximize}.cpp
// against the result.  With -DMB_GLUE_MOCK the reference headers are not pulled in (tests/cxx/mock_reference.h stands in).
#pragma once
#ifndef MB_GLUE_MOCK
#include "eval.h"      // EvaluatedMachine, Tokenizer           (src/eval.h:11-98)
#include "seqpair.h"   // SeqPair, SeqPairList, Envelope        (src/seqpair.h:56-116)
#include "machine.h"   // Machine, MachinePath, MachineTransition (src/machine.h)
#include "params.h"    // ParamAssign                           (src/params.h:26-30)
#include "weight.h"    // WeightAlgebra                         (src/weight.h:83-90)
#include "util.h"      // escaped_str                           (src/util.h:100)
#endif
#include "mb_dp.hpp"

namespace MachineBossHIP {
template <> struct PathOf<MachineBoss::Machine> { typedef MachineBoss::MachinePath type; };   // paths come back as the reference's MachinePath
}

namespace MachineBoss {

// what MachineCounts::paramCounts / writeParamCountsJson call (src/counts.cpp:80-106): the caller's own algebra
struct HipdpAlgebraPolicy : WeightAlgebra {
  static std::string escaped_str(const std::string &s) { return MachineBoss::escaped_str(s); }
};

typedef MachineBossHIP::ForwardMatrixT<EvaluatedMachine, SeqPair> ForwardMatrix;                         // src/forward.h:19-27
typedef MachineBossHIP::RollingOutputForwardMatrixT<EvaluatedMachine, SeqPair> RollingOutputForwardMatrix;   // src/forward.h:29
typedef MachineBossHIP::BackwardMatrixT<EvaluatedMachine, SeqPair> BackwardMatrix;                       // src/backward.h:10-59
typedef MachineBossHIP::ViterbiMatrixT<EvaluatedMachine, SeqPair> ViterbiMatrix;                         // src/viterbi.h:9-18
typedef MachineBossHIP::MachineCountsT<EvaluatedMachine, SeqPair, SeqPairList, HipdpAlgebraPolicy> MachineCounts;   // src/counts.h:11-25
template <class IndexMapper> using DPMatrix = MachineBossHIP::DPMatrixCore;                               // DPMatrix<IdentityIndexMapper>::TraceTerminator etc.
struct IdentityIndexMapper {};

}  // namespace MachineBoss

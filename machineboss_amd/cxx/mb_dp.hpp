// mb_dp.hpp -- C++ shim with the reference's DP class names on top of the C-ABI (include/mbhip.h).
//
// This is the code a Machine Boss maintainer would put behind src/{forward,backward,viterbi,counts}.h to run the DP
// hot path on an MI355X (see INTEGRATION.md for the glue to the real EvaluatedMachine / SeqPair / Machine classes).
// Same contract as the reference: construction is computation (src/forward.defs.h:1-21, viterbi.cpp:6-16,
// backward.cpp:6-16); results are read through logLike(), cell(), path(), getCounts()/MachineCounts::count; errors are
// std::runtime_error carrying the library message (the reference throws runtime_error("Abort"), src/util.cpp:39-48).
// Header-only; link with -lmbhip.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mbhip.h"

namespace MachineBossHIP {

typedef int InputToken;
typedef int OutputToken;
typedef unsigned long long StateIndex;   // src/machine.h:21

inline void check(int rc) { if (rc) throw std::runtime_error(mb_last_error()); }

// Flattened EvaluatedMachine (src/eval.h:59-98): struct-of-arrays over global transition ids
// e = transOffset[src] + transIndex, the order EvaluatedMachine::init visits them (src/eval.cpp:47-69).
struct FlatMachine {
  int nStates = 0, nInTok = 0, nOutTok = 0;          // alphabet sizes exclude epsilon (token 0)
  std::vector<uint32_t> src, dst, transIndex;
  std::vector<uint16_t> inTok, outTok;
  std::vector<double> logWeight;
  std::vector<size_t> transOffset;                    // [nStates+1]
  std::vector<size_t> outDegree;
  mutable mb_machine *dev = nullptr;

  size_t nTransitions() const { return src.size(); }
  StateIndex startState() const { return 0; }
  StateIndex endState() const { return nStates - 1; }

  // edges must be appended in ascending source state, in each state's transition-list order
  void addTransition(uint32_t s, uint32_t d, uint16_t in, uint16_t out, double lw) {
    if (outDegree.empty()) outDegree.assign(nStates, 0);
    transIndex.push_back((uint32_t)(outDegree[s]++));
    src.push_back(s); dst.push_back(d); inTok.push_back(in); outTok.push_back(out); logWeight.push_back(lw);
  }
  void finish() {      // transOffset = prefix sum of out-degrees (src/eval.cpp:65-68)
    if (outDegree.empty()) outDegree.assign(nStates, 0);
    transOffset.assign(nStates + 1, 0);
    for (int s = 0; s < nStates; ++s) transOffset[s + 1] = transOffset[s] + outDegree[s];
  }
  mb_machine *device() const {
    if (!dev) {
      dev = mb_machine_create(nStates, nInTok, nOutTok, (int64_t)src.size(), src.data(), dst.data(), inTok.data(),
                              outTok.data(), logWeight.data());
      if (!dev) throw std::runtime_error(mb_last_error());
    }
    return dev;
  }
  void setLogWeights(const std::vector<double> &lw) {     // per EM iteration (src/fitter.cpp:28-29)
    logWeight = lw;
    if (dev) check(mb_machine_set_weights(dev, logWeight.data()));
  }
  ~FlatMachine() { if (dev) mb_machine_destroy(dev); }
  FlatMachine() = default;
  FlatMachine(const FlatMachine &) = delete;
  FlatMachine &operator=(const FlatMachine &) = delete;
};

struct TokSeqPair {                       // a tokenised SeqPair (Tokenizer::tokenize, src/eval.h:29-41)
  std::vector<InputToken> input;
  std::vector<OutputToken> output;
};

struct PathStep { StateIndex src; size_t transIndex; };   // MachinePath as (state, index into its transition list)

// Envelope (src/seqpair.h:75-97): cell (x,y) exists <=> inStart[y] <= x < inEnd[y].  An alignment column is
// (gotInput, gotOutput); initPath / initPathArea restate src/seqpair.cpp:134-182.
struct Envelope {
  typedef std::pair<bool, bool> AlignCol;
  long inLen = 0, outLen = 0;
  std::vector<int32_t> inStart{0}, inEnd{1};
  void clear() { inLen = outLen = 0; inStart.assign(1, 0); inEnd.assign(1, 1); }
  void initFull(long il, long ol) { inLen = il; outLen = ol; inStart.assign(ol + 1, 0); inEnd.assign(ol + 1, (int32_t)il + 1); }
  void initPath(const std::vector<AlignCol> &cols) {
    clear();
    for (const AlignCol &c : cols) {
      if (!c.first && c.second) { inStart.push_back(inEnd.back() - 1); inEnd.push_back(inEnd.back()); ++outLen; }
      else if (c.first && !c.second) { ++inEnd.back(); ++inLen; }
      else if (c.first && c.second) { inStart.push_back(inEnd.back()); inEnd.push_back(inEnd.back() + 1); ++inLen; ++outLen; }
    }
  }
  void initPathArea(const std::vector<AlignCol> &cols, size_t width) {
    clear();
    std::vector<int32_t> match; std::vector<size_t> nBefore(1, 0);
    for (const AlignCol &c : cols) {
      if (c.first && c.second) match.push_back((int32_t)inLen);
      if (c.first) ++inLen;
      if (c.second) { ++outLen; nBefore.push_back(match.size()); }
    }
    inStart.clear(); inEnd.clear();
    for (long j = 0; j <= outLen; ++j) {
      int32_t iStart = 0, iEnd = (int32_t)inLen + 1;
      if (nBefore[j] > width) iStart = match[nBefore[j] - width - 1] + 1;
      if (match.size() - nBefore[j] > width) iEnd = match[nBefore[j] + width] + 1;
      inStart.push_back(iStart); inEnd.push_back(iEnd);
    }
  }
  bool contains(long x, long y) const { return y >= 0 && y <= outLen && x >= inStart[y] && x < inEnd[y]; }
  bool fits(const TokSeqPair &sp) const { return inLen == (long)sp.input.size() && outLen == (long)sp.output.size(); }
  bool isFull() const {
    for (long y = 0; y <= outLen; ++y) if (inStart[y] != 0 || inEnd[y] != inLen + 1) return false;
    return true;
  }
  std::vector<long long> offsets() const {     // src/seqpair.cpp:195-204
    std::vector<long long> r(1, 0);
    for (long y = 0; y <= outLen; ++y) r.push_back(r.back() + inEnd[y] - inStart[y]);
    return r;
  }
};

// DPMatrix<IdentityIndexMapper> (src/dpmatrix.h:64-163)
class DPMatrix {
protected:
  std::vector<double> cellStorage;
  void fill(int mode, int startState) {
    cellStorage.resize((size_t)(inLen + 1) * (outLen + 1) * nStates);
    if (!env.fits(seqPair)) throw std::runtime_error("Envelope/sequence mismatch");      // DPMatrix::alloc, src/dpmatrix.defs.h:31
    const bool full = env.isFull();
    check(mb_fill_env(machine.device(), mode, seqPair.input.data(), inLen, seqPair.output.data(), outLen, startState,
                      full ? nullptr : env.inStart.data(), full ? nullptr : env.inEnd.data(), cellStorage.data()));
  }
public:
  const FlatMachine &machine;
  const TokSeqPair &seqPair;
  const long inLen, outLen;
  const StateIndex nStates;
  Envelope env;     // the caller's glue passes Envelope(seqPair): path envelope of an aligned pair, else full (quirk Q1)
  DPMatrix(const FlatMachine &m, const TokSeqPair &sp)
      : machine(m), seqPair(sp), inLen((long)sp.input.size()), outLen((long)sp.output.size()), nStates(m.nStates) { env.initFull(inLen, outLen); }
  DPMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e)
      : machine(m), seqPair(sp), inLen((long)sp.input.size()), outLen((long)sp.output.size()), nStates(m.nStates), env(e) {}
  double cell(long inPos, long outPos, StateIndex state) const {
    if (inPos < 0 || inPos > inLen || outPos < 0 || outPos > outLen) return -std::numeric_limits<double>::infinity();
    return cellStorage[((size_t)outPos * (inLen + 1) + inPos) * nStates + state];
  }
  double startCell() const { return cell(0, 0, machine.startState()); }
  double endCell() const { return cell(inLen, outLen, machine.endState()); }
};

class ForwardMatrix : public DPMatrix {      // src/forward.h:19-27
public:
  ForwardMatrix(const FlatMachine &m, const TokSeqPair &sp, StateIndex startState = 0) : DPMatrix(m, sp) { fill(MB_FORWARD, (int)startState); }
  ForwardMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e, StateIndex startState = 0) : DPMatrix(m, sp, e) { fill(MB_FORWARD, (int)startState); }
  double logLike() const { return endCell(); }
};

class BackwardMatrix : public DPMatrix {     // src/backward.h:44-59
public:
  BackwardMatrix(const FlatMachine &m, const TokSeqPair &sp) : DPMatrix(m, sp) { fill(MB_BACKWARD, 0); }
  BackwardMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e) : DPMatrix(m, sp, e) { fill(MB_BACKWARD, 0); }
  double logLike() const { return startCell(); }
};

class ViterbiMatrix : public DPMatrix {      // src/viterbi.h:9-18
public:
  ViterbiMatrix(const FlatMachine &m, const TokSeqPair &sp) : DPMatrix(m, sp) { fill(MB_VITERBI, 0); }
  ViterbiMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e) : DPMatrix(m, sp, e) { fill(MB_VITERBI, 0); }
  double logLike() const { return endCell(); }
  std::vector<PathStep> path() const {       // traceBack (src/dpmatrix.defs.h:61-110), run on the device
    if (!(endCell() > -std::numeric_limits<double>::infinity())) throw std::runtime_error("Can't do traceback: no finite-weight paths");
    const int64_t inOff[2] = {0, inLen}, outOff[2] = {0, outLen};
    const int64_t cap = mb_viterbi_path_bound(machine.device(), inLen, outLen);
    std::vector<uint32_t> edges((size_t)cap);
    int64_t off[2] = {0, 0};
    double ll = 0;
    check(mb_viterbi_batch(machine.device(), 1, seqPair.input.data(), inOff, seqPair.output.data(), outOff, &ll, off, edges.data(), cap));
    std::vector<PathStep> p;
    for (int64_t k = 0; k < off[1]; ++k) p.push_back({machine.src[edges[k]], machine.transIndex[edges[k]]});
    return p;
  }
};

// RollingOutputForwardMatrix (src/forward.h:29, dpmatrix.h:46-58): log-likelihood only, no matrix in HBM
class RollingOutputForwardMatrix {
  double ll;
public:
  RollingOutputForwardMatrix(const FlatMachine &m, const TokSeqPair &sp) {
    const int64_t inOff[2] = {0, (int64_t)sp.input.size()}, outOff[2] = {0, (int64_t)sp.output.size()};
    check(mb_forward_batch(m.device(), 1, sp.input.data(), inOff, sp.output.data(), outOff, MB_ROLLING, &ll));
  }
  double logLike() const { return ll; }
};

// MachineCounts (src/counts.h:11-25): E-step over a list of pairs in ONE device call
struct MachineCounts {
  std::vector<std::vector<double>> count;    // count[state][transIndex]
  double loglike = 0;
  MachineCounts() = default;
  explicit MachineCounts(const FlatMachine &m) { init(m); }
  MachineCounts(const FlatMachine &m, const std::vector<TokSeqPair> &pairs) { init(m); add(m, pairs); }
  void init(const FlatMachine &m) {
    loglike = 0;
    count.assign(m.nStates, {});
    for (int s = 0; s < m.nStates; ++s) count[s].assign(m.transOffset[s + 1] - m.transOffset[s], 0.0);
  }
  // envelopes: empty = all full; otherwise one per pair (MachineCounts(eval, seqPairList, envelopes), src/counts.cpp:37-43)
  std::vector<double> add(const FlatMachine &m, const std::vector<TokSeqPair> &pairs, const std::vector<Envelope> &envelopes = {}) {
    std::vector<InputToken> in; std::vector<OutputToken> out;
    std::vector<int64_t> inOff(1, 0), outOff(1, 0);
    for (const TokSeqPair &sp : pairs) {
      in.insert(in.end(), sp.input.begin(), sp.input.end()); out.insert(out.end(), sp.output.begin(), sp.output.end());
      inOff.push_back((int64_t)in.size()); outOff.push_back((int64_t)out.size());
    }
    std::vector<double> flat(m.nTransitions(), 0.0), ll(pairs.size(), 0.0);
    double s = 0;
    mb_batch *b = mb_batch_create(m.device(), (int64_t)pairs.size(), in.data(), inOff.data(), out.data(), outOff.data());
    if (!b) throw std::runtime_error(mb_last_error());
    int rc = 0;
    if (!envelopes.empty()) {
      if (envelopes.size() != pairs.size()) { mb_batch_destroy(b); throw std::runtime_error("Envelope/training set mismatch"); }
      std::vector<int64_t> envOff(1, 0); std::vector<int32_t> st(1, 0), en(1, 0);
      st.clear(); en.clear();
      for (const Envelope &e : envelopes) {
        if (!e.isFull()) { st.insert(st.end(), e.inStart.begin(), e.inStart.end()); en.insert(en.end(), e.inEnd.begin(), e.inEnd.end()); }
        envOff.push_back((int64_t)st.size());
      }
      if (st.empty()) { st.push_back(0); en.push_back(0); }
      rc = mb_batch_set_envelopes(b, envOff.data(), st.data(), en.data());
    }
    if (!rc) rc = mb_batch_counts(b, flat.data(), &s, ll.data());
    mb_batch_destroy(b);
    check(rc);
    for (size_t e = 0; e < flat.size(); ++e) count[m.src[e]][m.transIndex[e]] += flat[e];
    loglike += s;
    return ll;
  }
  // operator+= over every rank of `comm` at once: one RCCL all-reduce of nTransitions + 1 doubles (mb_allreduce_counts);
  // comm == nullptr (single process) leaves the counts as they are
  void allReduce(mb_comm *comm) {
    std::vector<double> flat;
    for (const auto &row : count) flat.insert(flat.end(), row.begin(), row.end());
    check(mb_allreduce_counts(comm, flat.data(), flat.size(), &loglike));
    size_t k = 0;
    for (auto &row : count) for (double &c : row) c = flat[k++];
  }
  MachineCounts &operator+=(const MachineCounts &o) {       // src/counts.cpp:66-71 (the RCCL all-reduce across ranks)
    for (size_t s = 0; s < count.size(); ++s) for (size_t t = 0; t < count[s].size(); ++t) count[s][t] += o.count[s][t];
    loglike += o.loglike;
    return *this;
  }
};

}  // namespace MachineBossHIP

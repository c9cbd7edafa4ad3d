// mb_dp.hpp -- C++ shim with the reference's DP class interface on top of the C-ABI (include/mbhip.h).
//
// This is the code a Machine Boss maintainer puts behind src/{dpmatrix,forward,backward,viterbi,counts}.h to run the DP
// hot path on an MI355X.  Same contract as the reference: construction is computation (src/forward.defs.h:1-21,
// viterbi.cpp:6-16, backward.cpp:6-16); results are read through logLike(), cell(), path(), getCounts() /
// MachineCounts::count; errors are std::runtime_error carrying the library message (the reference throws
// runtime_error("Abort") after printing, src/util.cpp:39-48).
//
// What a constructor computes is what the reference's CALLERS read (target/boss.cpp:796-800,826-833, src/api.cpp:31-66):
//  * ViterbiMatrix: the score and the path, through the kernel family's traceback-byte / traceback-code sweep
//    (mb_batch_viterbi on the one pair) -- logLike() and path(machine) answer from those;
//  * ForwardMatrix: the log-likelihood, through the rolling sweep (mb_batch_forward, MB_ROLLING);
//  * BackwardMatrix: nothing yet.
// The fp64 MATRIX (8 bytes per cell over PCIe: 64 MB for a 1 kb x 1 kb dnapsw pair, 10.6 GB for a 487 aa x 10 kb psw2dna
// pair) is fetched with mb_fill_env LAZILY, by the first cell() / writeJson / selector-traceBack / getCounts(forward, ...)
// that needs it; matrixFills() counts those fetches.  prefetch(eval, seqPairs, what) runs ONE batched device call for a
// whole SeqPairList and lets the matrices constructed afterwards in the caller's unchanged `for (seqPair : data.seqPairs)`
// loop pick their results up.
//
// Two layers:
//  * neutral types -- FlatMachine (struct-of-arrays EvaluatedMachine), TokSeqPair, Envelope, and the classes DPMatrix /
//    ForwardMatrix / BackwardMatrix / ViterbiMatrix / RollingOutputForwardMatrix / MachineCounts over them.  The fills run
//    on the GPU; the functions that only WALK a finished matrix (traceBack x4, traceForward x3, selectMaxTrans,
//    randomTransSelector, ForwardMatrix::samplePath, BackwardMatrix::getCounts with a visitor / postTransQueue /
//    traceFrom x3, writeJson: src/dpmatrix.h:136-162, src/forward.h:24-25, src/backward.h:50-58) run on the host over the
//    device-filled matrix, candidate order and quirks as in the reference;
//  * class templates over the CALLER's types (ForwardMatrixT<EvaluatedMachine, SeqPair> ...), duck-typed on the members
//    the reference's classes have (eval.state[s].outgoing / logTransWeight / name, eval.inputTokenizer,
//    seqPair.input.seq / .name / alignment, machine.state[s].getTransition(ti), MachinePath::trans), with the
//    reference's constructor signatures and public members (machine, seqPair, input, output, inLen, outLen, nStates),
//    so that boss.cpp-style callers compile unchanged once the glue header typedefs them (INTEGRATION.md section 2;
//    tests/cxx/ compiles exactly that against a mock of the reference's headers).
// Header-only; link with -lmbhip.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <iomanip>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <ostream>
#include <queue>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mbhip.h"

namespace MachineBossHIP {

typedef int InputToken;
typedef int OutputToken;
typedef unsigned long long StateIndex;   // src/machine.h:21
typedef size_t TransIndex;               // EvaluatedMachineState::TransIndex, src/eval.h:60

inline void check(int rc) { if (rc) throw std::runtime_error(mb_last_error()); }
inline double negInf() { return -std::numeric_limits<double>::infinity(); }

// fp64 matrices fetched from the device so far by this process (mb_fill_env calls of the lazy DP classes): a caller -- or a
// test -- can tell that a loop of logLike() / path(machine) calls moved no matrix over PCIe
inline long &matrixFillCounter() { static long n = 0; return n; }
inline long matrixFills() { return matrixFillCounter(); }

// Results of one batched device call, waiting for the matrices a caller's loop constructs pair by pair (prefetch(), below).
// Keyed by the ADDRESS of the caller's SeqPair (the loop variable of `for (const auto& seqPair : data.seqPairs)` is a
// reference to the list element); an entry is used only if the pair at that address still tokenises to the same sequences
// and has the same envelope.
enum { PrefetchLogLike = 1, PrefetchViterbi = 2 };
struct PrefetchedPair {
  std::vector<int> in, out;
  uint64_t fingerprint = 0;         // of the caller's symbol sequences (template layer): the pair at this address is still the pair that was prefetched
  std::vector<int32_t> envStart, envEnd;
  double ll = 0;
  std::vector<uint32_t> edges;      // Viterbi path as global edge ids, start -> end
};
struct PrefetchStore {
  std::map<const void *, PrefetchedPair> loglike, viterbi;
  void clear() { loglike.clear(); viterbi.clear(); }
};

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
  // host copies of the reference's iteration orders (src/eval.h:66-68): CSR by (state, inTok, outTok) over edge ids,
  // rows sorted by the other endpoint, then by insertion order -- what the nested map / multimap iterate
  mutable std::vector<uint32_t> inEdge, outEdge;
  mutable std::vector<size_t> inOff, outOff;
  mutable PrefetchStore prefetched;                   // results of prefetch() for THESE weights
  // single-character alphabets (DNA, protein: every preset) tokenise through a 256-entry table instead of the caller's
  // std::map<string, Token> (src/eval.h:29-41: two map look-ups per symbol); anything else goes to the caller's tokenizer
  struct CharTable { bool usable = false; int tok[256]; };
  CharTable inChars, outChars;

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
    inEdge.clear(); outEdge.clear();
  }
  size_t key(StateIndex st, int it, int ot) const { return ((size_t)st * (nInTok + 1) + it) * (nOutTok + 1) + ot; }
  void buildOrders() const {
    if (!inOff.empty()) return;
    for (int incoming = 0; incoming < 2; ++incoming) {
      std::vector<uint32_t> &edge = incoming ? inEdge : outEdge;
      std::vector<size_t> &off = incoming ? inOff : outOff;
      const size_t nKeys = (size_t)nStates * (nInTok + 1) * (nOutTok + 1);
      edge.resize(nTransitions());
      for (size_t e = 0; e < edge.size(); ++e) edge[e] = (uint32_t)e;
      auto rowOf = [&](uint32_t e) { return key(incoming ? dst[e] : src[e], inTok[e], outTok[e]); };
      std::stable_sort(edge.begin(), edge.end(), [&](uint32_t a, uint32_t b) {
        const size_t ra = rowOf(a), rb = rowOf(b);
        if (ra != rb) return ra < rb;
        return (incoming ? src[a] : dst[a]) < (incoming ? src[b] : dst[b]);
      });
      off.assign(nKeys + 1, 0);
      for (uint32_t e : edge) off[rowOf(e) + 1]++;
      for (size_t k = 0; k < nKeys; ++k) off[k + 1] += off[k];
    }
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
    prefetched.clear();
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

struct PathStep { StateIndex src; TransIndex transIndex; };   // MachinePath as (state, index into its transition list)

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
  static bool overlapping(long s1, long e1, long s2, long e2) { return !(s1 >= e2 || s2 >= e1); }
  bool connected() const {                       // src/seqpair.cpp:188-193
    if ((long)inStart.size() != outLen + 1 || (long)inEnd.size() != outLen + 1) return false;
    bool conn = overlapping(inStart[0], inEnd[0], 0, 1);
    for (long y = 1; conn && y <= outLen; ++y) conn = conn && overlapping(inStart[y - 1], inEnd[y - 1] + 1, inStart[y], inEnd[y]);
    return conn && overlapping(inStart[outLen], inEnd[outLen], inLen, inLen + 1);
  }
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

// A tokenised SeqPairList resident on the device (mb_batch), with the envelopes of its pairs: what every class below runs its
// sweeps on, one pair or a whole list alike.  envelopes: empty = all full; otherwise one per pair.
class DeviceBatch {
  mb_batch *b = nullptr;
  const FlatMachine &flat;
  std::vector<int64_t> bound;
public:
  DeviceBatch(const FlatMachine &m, const std::vector<const TokSeqPair *> &pairs, const std::vector<const Envelope *> &envelopes) : flat(m) {
    std::vector<InputToken> in; std::vector<OutputToken> out;
    std::vector<int64_t> inOff(1, 0), outOff(1, 0);
    for (const TokSeqPair *sp : pairs) {
      in.insert(in.end(), sp->input.begin(), sp->input.end()); out.insert(out.end(), sp->output.begin(), sp->output.end());
      inOff.push_back((int64_t)in.size()); outOff.push_back((int64_t)out.size());
      bound.push_back(mb_viterbi_path_bound(m.device(), (int64_t)sp->input.size(), (int64_t)sp->output.size()));
    }
    if (in.empty()) in.push_back(0);
    if (out.empty()) out.push_back(0);
    b = mb_batch_create(m.device(), (int64_t)pairs.size(), in.data(), inOff.data(), out.data(), outOff.data());
    if (!b) throw std::runtime_error(mb_last_error());
    if (!envelopes.empty()) {
      if (envelopes.size() != pairs.size()) { mb_batch_destroy(b); throw std::runtime_error("Envelope/training set mismatch"); }
      std::vector<int64_t> envOff(1, 0); std::vector<int32_t> st, en;
      for (const Envelope *e : envelopes) {
        if (!e->isFull()) { st.insert(st.end(), e->inStart.begin(), e->inStart.end()); en.insert(en.end(), e->inEnd.begin(), e->inEnd.end()); }
        envOff.push_back((int64_t)st.size());
      }
      if (!st.empty() && mb_batch_set_envelopes(b, envOff.data(), st.data(), en.data())) { mb_batch_destroy(b); throw std::runtime_error(mb_last_error()); }
    }
  }
  DeviceBatch(const DeviceBatch &) = delete;
  DeviceBatch &operator=(const DeviceBatch &) = delete;
  ~DeviceBatch() { if (b) mb_batch_destroy(b); }
  size_t size() const { return bound.size(); }
  std::vector<double> forward(int flags) { std::vector<double> ll(size(), 0.0); check(mb_batch_forward(b, flags, ll.data())); return ll; }
  // scores, and the paths as global edge ids: pair p owns edges[off[p] .. off[p+1])
  void viterbi(std::vector<double> &ll, std::vector<int64_t> &off, std::vector<uint32_t> &edges) {
    int64_t cap = 0;
    for (int64_t n : bound) cap += n;
    ll.assign(size(), 0.0); off.assign(size() + 1, 0); edges.resize((size_t)std::max<int64_t>(cap, 1));
    check(mb_batch_viterbi(b, ll.data(), off.data(), edges.data(), cap));
    edges.resize((size_t)off[size()]);
  }
  std::vector<double> counts(std::vector<double> &flatCounts, double &loglikeSum) {
    std::vector<double> ll(size(), 0.0);
    check(mb_batch_counts(b, flatCounts.data(), &loglikeSum, ll.data()));
    return ll;
  }
};

// A path as the caller's transition objects; PathOf<MachineT> lets the glue substitute the reference's MachinePath
// (anything with a `trans` sequence supporting push_back / push_front and a concatenate()).
template <class TransT>
struct MachinePathT {
  std::list<TransT> trans;
  MachinePathT() {}
  explicit MachinePathT(const TransT &t) : trans(1, t) {}
  void clear() { trans.clear(); }
  MachinePathT concatenate(const MachinePathT &o) const { MachinePathT r(*this); r.trans.insert(r.trans.end(), o.trans.begin(), o.trans.end()); return r; }
};
template <class MachineT>
struct PathOf {
  typedef typename std::decay<decltype(std::declval<const MachineT &>().state[0].getTransition(0))>::type Transition;
  typedef MachinePathT<Transition> type;
};

// a result of prefetch() for the (already tokenised) pair at `key`, if it is still about these sequences and this envelope
inline const PrefetchedPair *prefetchedFor(const std::map<const void *, PrefetchedPair> &store, const void *key, const std::vector<InputToken> &in,
                                           const std::vector<OutputToken> &out, const Envelope &env) {
  if (!key || store.empty()) return nullptr;
  const auto it = store.find(key);
  if (it == store.end()) return nullptr;
  const PrefetchedPair &pp = it->second;
  const bool full = env.isFull();
  if (pp.in != in || pp.out != out || (full ? !pp.envStart.empty() : (pp.envStart != env.inStart || pp.envEnd != env.inEnd))) return nullptr;
  return &pp;
}

// ---- DPMatrix<IdentityIndexMapper> (src/dpmatrix.h:64-163) over neutral types -----------------------------------------------
class DPMatrixCore {
public:
  typedef long InputIndex;
  typedef long OutputIndex;
  typedef std::function<bool(InputIndex, OutputIndex, StateIndex, TransIndex)> TraceTerminator;
  typedef std::function<void(StateIndex, TransIndex, double)> TransVisitor;
  typedef std::function<size_t(const std::vector<double> &)> TransSelector;

protected:
  mutable std::vector<double> cellStorage;      // the fp64 matrix, fetched from the device by the first reader (ensureMatrix)
  mutable bool haveMatrix = false;
  const FlatMachine &flat;
  const int fillMode;                           // MB_FORWARD / MB_VITERBI / MB_BACKWARD
  const int fillStartState;
  // DPMatrix::alloc's assertions (src/dpmatrix.defs.h:31-32) hold at construction, whether or not a matrix is ever fetched
  void checkEnvelope() const {
    if (env.inLen != inLen || env.outLen != outLen) throw std::runtime_error("Envelope/sequence mismatch");
    if (!env.connected()) throw std::runtime_error("Envelope is not connected");
  }
  void ensureMatrix() const {
    if (haveMatrix) return;
    cellStorage.resize((size_t)(inLen + 1) * (outLen + 1) * nStates);
    const bool full = env.isFull();
    check(mb_fill_env(flat.device(), fillMode, input.data(), inLen, output.data(), outLen, fillStartState,
                      full ? nullptr : env.inStart.data(), full ? nullptr : env.inEnd.data(), cellStorage.data()));
    ++matrixFillCounter();
    haveMatrix = true;
  }
  // the one pair as a device batch (scores, paths and log-likelihoods without the matrix)
  std::unique_ptr<DeviceBatch> onePairBatch() const {
    pairView.input = input; pairView.output = output;
    std::vector<const TokSeqPair *> ps(1, &pairView);
    std::vector<const Envelope *> es;
    if (!env.isFull()) es.push_back(&env);
    return std::unique_ptr<DeviceBatch>(new DeviceBatch(flat, ps, es));
  }
private:
  mutable TokSeqPair pairView;
protected:
  // DPMatrix::iterate over one label group (src/dpmatrix.h:101-115): candidates in multimap order
  void pathIterate(const TransVisitor &visit, bool incoming, StateIndex state, InputToken inTok, OutputToken outTok, InputIndex inPos, OutputIndex outPos) const {
    flat.buildOrders();
    const std::vector<size_t> &off = incoming ? flat.inOff : flat.outOff;
    const std::vector<uint32_t> &edge = incoming ? flat.inEdge : flat.outEdge;
    const size_t k = flat.key(state, inTok, outTok);
    for (size_t a = off[k]; a < off[k + 1]; ++a) {
      const uint32_t e = edge[a];
      const StateIndex other = incoming ? flat.src[e] : flat.dst[e];
      visit(other, flat.transIndex[e], cell(inPos, outPos, other) + flat.logWeight[e]);
    }
  }

public:
  const std::vector<InputToken> input;
  const std::vector<OutputToken> output;
  const InputIndex inLen;
  const OutputIndex outLen;
  const StateIndex nStates;
  Envelope env;     // Envelope(seqPair): the path envelope of an aligned pair, else full (quirk Q1: src/dpmatrix.defs.h:16-17)

  DPMatrixCore(const FlatMachine &m, const std::vector<InputToken> &in, const std::vector<OutputToken> &out, const Envelope &e, int mode = MB_FORWARD, int startState = 0)
      : flat(m), fillMode(mode), fillStartState(startState), input(in), output(out), inLen((long)in.size()), outLen((long)out.size()), nStates(m.nStates), env(e) { checkEnvelope(); }
  bool matrixFetched() const { return haveMatrix; }

  const FlatMachine &flatMachine() const { return flat; }
  // the const accessor of the reference: -inf outside the envelope (src/dpmatrix.h:142-144)
  double cell(InputIndex inPos, OutputIndex outPos, StateIndex state) const {
    if (!env.contains(inPos, outPos)) return negInf();
    ensureMatrix();
    return cellStorage[((size_t)outPos * (inLen + 1) + inPos) * nStates + state];
  }
  double startCell() const { return cell(0, 0, flat.startState()); }
  double endCell() const { return cell(inLen, outLen, flat.endState()); }

  // DPMatrix::writeJson (src/dpmatrix.defs.h:39-53): every cell at setprecision(5), input position outermost.
  // nameOf(s) must print the state's name as JSON (the reference streams machine.state[s].name, a json value).
  template <class NameFn>
  void writeJsonWith(std::ostream &outs, const std::string &inputName, const std::string &outputName, NameFn nameOf) const {
    outs << "{" << std::endl << " \"input\": \"" << inputName << "\"," << std::endl << " \"output\": \"" << outputName << "\"," << std::endl << " \"cell\": [";
    for (InputIndex i = 0; i <= inLen; ++i)
      for (OutputIndex o = 0; o <= outLen; ++o)
        for (StateIndex s = 0; s < nStates; ++s) {
          outs << ((i || o || s) ? "," : "") << std::endl << "  { \"inPos\": " << i << ", \"outPos\": " << o << ", \"state\": ";
          nameOf(outs, s);
          outs << ", \"logLike\": " << std::setprecision(5) << cell(i, o, s) << " }";
        }
    outs << std::endl << " ]" << std::endl << "}" << std::endl;
  }

  static TransVisitor addTransToTraceOptions(std::vector<StateIndex> &state, std::vector<TransIndex> &transIndex, std::vector<double> &loglike) {
    return [&](StateIndex s, TransIndex ti, double tll) { state.push_back(s); transIndex.push_back(ti); loglike.push_back(tll); };
  }
  // std::max_element: the FIRST maximum (src/dpmatrix.defs.h:171-174)
  static size_t selectMaxTrans(const std::vector<double> &logWeights) {
    return (size_t)std::distance(logWeights.begin(), std::max_element(logWeights.begin(), logWeights.end()));
  }
  // random_index over exp(logWeights) with random_double(rng) (src/dpmatrix.defs.h:176-186, src/util.h:102-106,151-165)
  template <class Generator>
  static TransSelector randomTransSelector(Generator &rng) {
    return [&rng](const std::vector<double> &logWeights) -> size_t {
      std::vector<double> weights;
      weights.reserve(logWeights.size());
      for (const double lw : logWeights) weights.push_back(std::exp(lw));
      double norm = 0;
      for (const double w : weights) { if (!(w >= 0)) throw std::runtime_error("Negative weights in random_index"); norm += w; }
      if (!(norm > 0)) throw std::runtime_error("Zero weights in random_index");
      double variate = (rng() / (((double)std::numeric_limits<typename Generator::result_type>::max()) + 1)) * norm;
      for (size_t n = 0; n < weights.size(); ++n)
        if ((variate -= weights[n]) <= 0) return n;
      return weights.size();
    };
  }

  // ---- the walkers proper: positions and (state, transIndex) only; the MachinePath forms below wrap them -----------------
  // DPMatrix::traceBack (m, inPos, outPos, s, stopTrace, selectTrans), src/dpmatrix.defs.h:82-110
  void traceBackSteps(InputIndex inPos, OutputIndex outPos, StateIndex s, const TraceTerminator &stopTrace, const TransSelector &selectTrans) const {
    if (!(cell(inPos, outPos, s) > negInf())) throw std::runtime_error("Can't do traceback: no finite-weight paths");
    while (inPos > 0 || outPos > 0 || s != 0) {
      std::vector<double> loglike; std::vector<StateIndex> source; std::vector<TransIndex> transIndex;
      const TransVisitor tv = addTransToTraceOptions(source, transIndex, loglike);
      const InputToken inTok = inPos ? input[inPos - 1] : 0;
      const OutputToken outTok = outPos ? output[outPos - 1] : 0;
      if (inPos && outPos) pathIterate(tv, true, s, inTok, outTok, inPos - 1, outPos - 1);
      if (inPos) pathIterate(tv, true, s, inTok, 0, inPos - 1, outPos);
      if (outPos) pathIterate(tv, true, s, 0, outTok, inPos, outPos - 1);
      pathIterate(tv, true, s, 0, 0, inPos, outPos);
      if (loglike.empty()) throw std::runtime_error("Traceback reached a cell without incoming transitions");   // (undefined behaviour in the reference)
      const size_t best = selectTrans(loglike);
      const StateIndex bestSource = source[best];
      const TransIndex bestTransIndex = transIndex[best];
      const size_t e = flat.transOffset[bestSource] + bestTransIndex;     // = m.state[bestSource].getTransition(bestTransIndex)
      if (flat.inTok[e]) --inPos;
      if (flat.outTok[e]) --outPos;
      s = bestSource;
      if (stopTrace(inPos, outPos, s, bestTransIndex)) break;
    }
  }
  // DPMatrix::traceForward (m, inPos, outPos, s, stopTrace, selectTrans), src/dpmatrix.defs.h:128-159
  void traceForwardSteps(InputIndex inPos, OutputIndex outPos, StateIndex s, const TraceTerminator &stopTrace, const TransSelector &selectTrans) const {
    if (!(cell(inPos, outPos, s) > negInf())) throw std::runtime_error("Can't do traceforward: no finite-weight paths");
    while (inPos < inLen || outPos < outLen || s != nStates - 1) {
      std::vector<double> loglike; std::vector<StateIndex> dest; std::vector<TransIndex> transIndex;
      const TransVisitor tv = addTransToTraceOptions(dest, transIndex, loglike);
      const bool endOfInput = (inPos == inLen), endOfOutput = (outPos == outLen);
      const InputToken inTok = endOfInput ? 0 : input[inPos];
      const OutputToken outTok = endOfOutput ? 0 : output[outPos];
      if (!endOfInput && !endOfOutput) pathIterate(tv, false, s, inTok, outTok, inPos + 1, outPos + 1);
      if (!endOfInput) pathIterate(tv, false, s, inTok, 0, inPos + 1, outPos);
      if (!endOfOutput) pathIterate(tv, false, s, 0, outTok, inPos, outPos + 1);
      pathIterate(tv, false, s, 0, 0, inPos, outPos);
      if (loglike.empty()) throw std::runtime_error("Traceforward reached a cell without outgoing transitions");
      const size_t best = selectTrans(loglike);
      const StateIndex bestDest = dest[best];
      const TransIndex bestTransIndex = transIndex[best];
      if (stopTrace(inPos, outPos, s, bestTransIndex)) break;
      const size_t e = flat.transOffset[s] + bestTransIndex;
      if (flat.dst[e] != bestDest) throw std::runtime_error("Traceforward error");
      if (flat.inTok[e]) ++inPos;
      if (flat.outTok[e]) ++outPos;
      s = bestDest;
    }
  }

  // ---- the reference's overloads (src/dpmatrix.h:150-162).  Quirk Q2 is kept: the MachinePath overloads that take a
  //      position ignore it and start at (inLen, outLen) (src/dpmatrix.defs.h:72-80,118-126); traceForward(m, selector)
  //      is traceBack(m, 0, 0, 0, selector) (:112-115).  Only the TraceTerminator overloads honour their position. -----
  template <class MachineT>
  typename PathOf<MachineT>::type traceBack(const MachineT &m, TransSelector ts = selectMaxTrans) const { return traceBack(m, inLen, outLen, nStates - 1, ts); }
  template <class MachineT>
  typename PathOf<MachineT>::type traceBack(const MachineT &m, StateIndex s, TransSelector ts = selectMaxTrans) const { return traceBack(m, inLen, outLen, s, ts); }
  template <class MachineT>
  typename PathOf<MachineT>::type traceBack(const MachineT &m, InputIndex, OutputIndex, StateIndex s, TransSelector ts = selectMaxTrans) const {
    typename PathOf<MachineT>::type path;
    const TraceTerminator stopTrace = [&](InputIndex, OutputIndex, StateIndex src, TransIndex ti) { path.trans.push_front(m.state[src].getTransition(ti)); return false; };
    traceBackSteps(inLen, outLen, s, stopTrace, ts);
    return path;
  }
  template <class MachineT>
  void traceBack(const MachineT &, InputIndex inPos, OutputIndex outPos, StateIndex s, TraceTerminator stopTrace, TransSelector ts = selectMaxTrans) const {
    traceBackSteps(inPos, outPos, s, stopTrace, ts);
  }
  template <class MachineT>
  typename PathOf<MachineT>::type traceForward(const MachineT &m, TransSelector ts = selectMaxTrans) const { return traceBack(m, 0, 0, 0, ts); }
  template <class MachineT>
  typename PathOf<MachineT>::type traceForward(const MachineT &m, InputIndex, OutputIndex, StateIndex s, TransSelector ts = selectMaxTrans) const {
    typename PathOf<MachineT>::type path;
    const TraceTerminator stopTrace = [&](InputIndex, OutputIndex, StateIndex src, TransIndex ti) { path.trans.push_back(m.state[src].getTransition(ti)); return false; };
    traceForwardSteps(inLen, outLen, s, stopTrace, ts);
    return path;
  }
  template <class MachineT>
  void traceForward(const MachineT &, InputIndex inPos, OutputIndex outPos, StateIndex s, TraceTerminator stopTrace, TransSelector ts = selectMaxTrans) const {
    traceForwardSteps(inPos, outPos, s, stopTrace, ts);
  }
};

class ForwardCore : public DPMatrixCore {      // src/forward.h:19-27
  mutable double ll = 0;
  mutable bool haveLL = false;
  const bool rollable;
public:
  // pre: this pair's result of prefetch(), if there is one.  NOTHING is swept at construction (a caller that builds a
  // ForwardMatrix only to walk it -- getCounts(forward, backward), samplePath -- paid a rolling sweep per pair for a value it never read)
  ForwardCore(const FlatMachine &m, const std::vector<InputToken> &in, const std::vector<OutputToken> &out, const Envelope &e, StateIndex startState = 0, const PrefetchedPair *pre = nullptr)
      : DPMatrixCore(m, in, out, e, MB_FORWARD, (int)startState), rollable(startState == 0) {      // (a caller-chosen start state, src/forward.h:24, exists on the matrix route only)
    if (rollable && pre) { ll = pre->ll; haveLL = true; }
  }
  // src/forward.defs.h:51-55.  Once the matrix has been fetched this IS its end cell, so that posteriors normalised with logLike()
  // agree with the cells they are computed from; before that, the rolling sweep's value (run on the first call; the two agree to
  // ~1e-9 relative when the sweep sums in another order than the materialised fill)
  double logLike() const {
    if (matrixFetched() || !rollable) return endCell();
    if (!haveLL) { ll = onePairBatch()->forward(MB_ROLLING)[0]; haveLL = true; }
    return ll;
  }
  // stochastic traceback (src/forward.cpp:17-23)
  template <class MachineT, class Generator>
  typename PathOf<MachineT>::type samplePath(const MachineT &m, Generator &rng) const { return traceBack(m, randomTransSelector(rng)); }
  template <class MachineT, class Generator>
  typename PathOf<MachineT>::type samplePath(const MachineT &m, StateIndex s, Generator &rng) const { return traceBack(m, s, randomTransSelector(rng)); }
};

class ViterbiCore : public DPMatrixCore {      // src/viterbi.h:9-18
  double score = 0;
  std::vector<uint32_t> pathEdges;             // the fill's own arg-max chain, traced on the device: global edge ids, start -> end
public:
  ViterbiCore(const FlatMachine &m, const std::vector<InputToken> &in, const std::vector<OutputToken> &out, const Envelope &e, const PrefetchedPair *pre = nullptr)
      : DPMatrixCore(m, in, out, e, MB_VITERBI, 0) {
    if (pre) { score = pre->ll; pathEdges = pre->edges; return; }
    std::vector<double> l; std::vector<int64_t> off;
    onePairBatch()->viterbi(l, off, pathEdges);
    score = l[0];
  }
  double logLike() const { return score; }     // = endCell(), bit for bit (the max semiring is exact in every kernel family)
  // src/viterbi.cpp:49-51: traceBack(m) with selectMaxTrans -- the first maximum in the reference's enumeration order is what
  // the device's traceback bytes / codes record, so the path is answered without the matrix
  template <class MachineT>
  typename PathOf<MachineT>::type path(const MachineT &m) const {
    typename PathOf<MachineT>::type p;
    for (const PathStep &st : path()) p.trans.push_back(m.state[st.src].getTransition(st.transIndex));
    return p;
  }
  std::vector<PathStep> path() const {         // as (state, transIndex) steps
    if (!(score > negInf())) throw std::runtime_error("Can't do traceback: no finite-weight paths");      // src/dpmatrix.defs.h:84
    std::vector<PathStep> p;
    p.reserve(pathEdges.size());
    for (const uint32_t e : pathEdges) p.push_back({flat.src[e], flat.transIndex[e]});
    return p;
  }
  const std::vector<uint32_t> &pathEdgeIds() const { return pathEdges; }
};

class BackwardCore : public DPMatrixCore {     // src/backward.h:10-59
public:
  typedef std::function<void(StateIndex, TransIndex, InputIndex, OutputIndex, double)> BackTransVisitor;
  struct PostTrans {
    InputIndex inPos; OutputIndex outPos; StateIndex src; TransIndex transIndex; double weight;
    bool operator<(const PostTrans &ptq) const { return weight < ptq.weight; }
  };
  typedef std::priority_queue<PostTrans> PostTransQueue;
  static BackTransVisitor transitionSorter(PostTransQueue &ptq) {
    return [&](StateIndex s, TransIndex ti, InputIndex ip, OutputIndex op, double postProb) { ptq.push(PostTrans{ip, op, s, ti, postProb}); };
  }
  BackwardCore(const FlatMachine &m, const std::vector<InputToken> &in, const std::vector<OutputToken> &out, const Envelope &e, const PrefetchedPair * = nullptr)
      : DPMatrixCore(m, in, out, e, MB_BACKWARD, 0) {}
  double logLike() const { return startCell(); }      // fetches the matrix: the reference's callers of BackwardMatrix all walk it (src/counts.cpp:57-61 is MachineCounts here)

  // BackwardMatrix::getCounts with a visitor (src/backward.cpp:62-87): every cell, every outgoing transition, in the
  // reference's order; the position handed to the visitor is the transition's DESTINATION cell (:77-83)
  void getCounts(const ForwardCore &forward, const BackTransVisitor &transCount) const {
    const double ll = logLike();
    for (OutputIndex outPos = outLen; outPos >= 0; --outPos) {
      const bool endOfOutput = (outPos == outLen);
      const OutputToken outTok = endOfOutput ? 0 : output[outPos];
      for (InputIndex inPos = env.inEnd[outPos] - 1; inPos >= env.inStart[outPos]; --inPos) {
        const bool endOfInput = (inPos == inLen);
        const InputToken inTok = endOfInput ? 0 : input[inPos];
        for (long long s = (long long)nStates - 1; s >= 0; --s) {
          const double logOddsRatio = forward.cell(inPos, outPos, (StateIndex)s) - ll;
          auto acc = [&](InputToken it, OutputToken ot, InputIndex ip, OutputIndex op) {
            pathIterate([&](StateIndex, TransIndex ti, double tll) { transCount((StateIndex)s, ti, ip, op, std::exp(logOddsRatio + tll)); }, false, (StateIndex)s, it, ot, ip, op);
          };
          if (!endOfInput && !endOfOutput) acc(inTok, outTok, inPos + 1, outPos + 1);
          if (!endOfInput) acc(inTok, 0, inPos + 1, outPos);
          if (!endOfOutput) acc(0, outTok, inPos, outPos + 1);
          acc(0, 0, inPos, outPos);
        }
      }
    }
  }
  PostTransQueue postTransQueue(const ForwardCore &forward) const {     // src/backward.cpp:52-56
    PostTransQueue ptq;
    getCounts(forward, transitionSorter(ptq));
    return ptq;
  }
  // traceFrom (src/backward.cpp:89-108)
  template <class MachineT>
  typename PathOf<MachineT>::type traceFrom(const MachineT &m, const ForwardCore &forward, InputIndex inPos, OutputIndex outPos, StateIndex state) const {
    return forward.traceBack(m, inPos, outPos, state).concatenate(traceForward(m, inPos, outPos, state));
  }
  template <class MachineT>
  typename PathOf<MachineT>::type traceFrom(const MachineT &m, const ForwardCore &forward, InputIndex inPos, OutputIndex outPos, StateIndex state, TransIndex transIndex) const {
    typedef typename PathOf<MachineT>::type Path;
    return forward.traceBack(m, inPos, outPos, state).concatenate(Path(m.state[state].getTransition(transIndex)).concatenate(traceForward(m, inPos, outPos, state)));
  }
  template <class MachineT>
  void traceFrom(const MachineT &m, const ForwardCore &forward, InputIndex inPos, OutputIndex outPos, StateIndex state, TransIndex transIndex, TraceTerminator stopTrace) const {
    if (!stopTrace(inPos, outPos, state, transIndex)) {
      forward.traceBack(m, inPos, outPos, state, stopTrace);
      const size_t e = flat.transOffset[state] + transIndex;
      const InputIndex nextInPos = inPos + (flat.inTok[e] ? 1 : 0);
      const OutputIndex nextOutPos = outPos + (flat.outTok[e] ? 1 : 0);
      traceForward(m, nextInPos, nextOutPos, (StateIndex)flat.dst[e], stopTrace);
    }
  }
};

// ---- neutral-type classes (FlatMachine + TokSeqPair), as before ------------------------------------------------------------
inline Envelope fullEnvelope(const TokSeqPair &sp) { Envelope e; e.initFull((long)sp.input.size(), (long)sp.output.size()); return e; }

class ForwardMatrix : public ForwardCore {
public:
  const FlatMachine &machine; const TokSeqPair &seqPair;
  ForwardMatrix(const FlatMachine &m, const TokSeqPair &sp, StateIndex startState = 0) : ForwardCore(m, sp.input, sp.output, fullEnvelope(sp), startState, prefetchedFor(m.prefetched.loglike, &sp, sp.input, sp.output, fullEnvelope(sp))), machine(m), seqPair(sp) {}
  ForwardMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e, StateIndex startState = 0) : ForwardCore(m, sp.input, sp.output, e, startState, prefetchedFor(m.prefetched.loglike, &sp, sp.input, sp.output, e)), machine(m), seqPair(sp) {}
};
class BackwardMatrix : public BackwardCore {
public:
  const FlatMachine &machine; const TokSeqPair &seqPair;
  BackwardMatrix(const FlatMachine &m, const TokSeqPair &sp) : BackwardCore(m, sp.input, sp.output, fullEnvelope(sp)), machine(m), seqPair(sp) {}
  BackwardMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e) : BackwardCore(m, sp.input, sp.output, e), machine(m), seqPair(sp) {}
};
class ViterbiMatrix : public ViterbiCore {
public:
  const FlatMachine &machine; const TokSeqPair &seqPair;
  ViterbiMatrix(const FlatMachine &m, const TokSeqPair &sp) : ViterbiCore(m, sp.input, sp.output, fullEnvelope(sp), prefetchedFor(m.prefetched.viterbi, &sp, sp.input, sp.output, fullEnvelope(sp))), machine(m), seqPair(sp) {}
  ViterbiMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope &e) : ViterbiCore(m, sp.input, sp.output, e, prefetchedFor(m.prefetched.viterbi, &sp, sp.input, sp.output, e)), machine(m), seqPair(sp) {}
};

// RollingOutputForwardMatrix (src/forward.h:29, dpmatrix.h:46-58): log-likelihood only, no matrix in HBM
class RollingOutputForwardMatrix {
  double ll;
public:
  RollingOutputForwardMatrix(const FlatMachine &m, const TokSeqPair &sp, const Envelope *env = nullptr) {
    const bool full = !env || env->isFull();
    if (const PrefetchedPair *pp = prefetchedFor(m.prefetched.loglike, &sp, sp.input, sp.output, env ? *env : fullEnvelope(sp))) { ll = pp->ll; return; }
    std::vector<const TokSeqPair *> ps(1, &sp);
    std::vector<const Envelope *> es;
    if (!full) es.push_back(env);
    ll = DeviceBatch(m, ps, es).forward(MB_ROLLING)[0];
  }
  double logLike() const { return ll; }
};

// prefetch over neutral types: ONE batched device call for the whole list; keys[p] = the address the matrices of pair p will
// be constructed from (see the template form below, which is what a boss.cpp-style caller uses)
inline void prefetchPairs(const FlatMachine &m, const std::vector<const TokSeqPair *> &pairs, const std::vector<const Envelope *> &envelopes,
                          const std::vector<const void *> &keys, int what, const std::vector<uint64_t> *fingerprints = nullptr) {
  m.prefetched.clear();
  if (pairs.empty()) return;
  bool anyEnv = false;
  for (const Envelope *e : envelopes) anyEnv = anyEnv || !e->isFull();
  DeviceBatch batch(m, pairs, anyEnv ? envelopes : std::vector<const Envelope *>());
  auto entry = [&](size_t p) {
    PrefetchedPair pp;
    pp.in = pairs[p]->input; pp.out = pairs[p]->output;
    if (fingerprints) pp.fingerprint = (*fingerprints)[p];
    if (!envelopes.empty() && !envelopes[p]->isFull()) { pp.envStart = envelopes[p]->inStart; pp.envEnd = envelopes[p]->inEnd; }
    return pp;
  };
  if (what & PrefetchLogLike) {
    const std::vector<double> ll = batch.forward(MB_ROLLING);
    for (size_t p = 0; p < pairs.size(); ++p) { PrefetchedPair pp = entry(p); pp.ll = ll[p]; m.prefetched.loglike[keys[p]] = std::move(pp); }
  }
  if (what & PrefetchViterbi) {
    std::vector<double> ll; std::vector<int64_t> off; std::vector<uint32_t> edges;
    batch.viterbi(ll, off, edges);
    for (size_t p = 0; p < pairs.size(); ++p) {
      PrefetchedPair pp = entry(p);
      pp.ll = ll[p]; pp.edges.assign(edges.begin() + off[p], edges.begin() + off[p + 1]);
      m.prefetched.viterbi[keys[p]] = std::move(pp);
    }
  }
}

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
    if (!envelopes.empty() && envelopes.size() != pairs.size()) throw std::runtime_error("Envelope/training set mismatch");
    std::vector<const TokSeqPair *> ps; std::vector<const Envelope *> es;
    for (const TokSeqPair &sp : pairs) ps.push_back(&sp);
    for (const Envelope &e : envelopes) es.push_back(&e);
    std::vector<double> flat(m.nTransitions(), 0.0);
    double s = 0;
    const std::vector<double> ll = DeviceBatch(m, ps, es).counts(flat, s);
    for (size_t e = 0; e < flat.size(); ++e) count[m.src[e]][m.transIndex[e]] += flat[e];
    loglike += s;
    return ll;
  }
  // BackwardMatrix::transitionCounter (src/backward.h:12-18)
  BackwardCore::BackTransVisitor transitionCounter() {
    return [this](StateIndex s, TransIndex ti, long, long, double postProb) { count[s][ti] += postProb; };
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
  MachineCounts &operator+=(const MachineCounts &o) {       // src/counts.cpp:66-71: counts only, like the reference
    for (size_t s = 0; s < count.size(); ++s) for (size_t t = 0; t < count[s].size(); ++t) count[s][t] += o.count[s][t];
    return *this;
  }
};

// ---- the caller's own types: duck-typed flattening, tokenising and envelopes --------------------------------------------------
// EvaluatedMachine-shaped (src/eval.h:59-98): state[s].{outgoing[in][out] -> multimap<dest, {logWeight, transIndex}>,
// logTransWeight, name}, inputTokenizer / outputTokenizer {tok2sym, tokenize()}.
template <class EvalT>
void flattenEvaluated(const EvalT &eval, FlatMachine &flat) {
  flat.nStates = (int)eval.state.size();
  flat.nInTok = (int)eval.inputTokenizer.tok2sym.size() - 1;      // token 0 = epsilon (src/eval.h:17)
  flat.nOutTok = (int)eval.outputTokenizer.tok2sym.size() - 1;
  for (StateIndex s = 0; s < eval.state.size(); ++s) {
    const size_t n = eval.state[s].logTransWeight.size();
    std::vector<uint32_t> dest(n); std::vector<uint16_t> in(n), out(n);
    for (const auto &iost : eval.state[s].outgoing)
      for (const auto &ost : iost.second)
        for (const auto &st : ost.second) { dest[st.second.transIndex] = (uint32_t)st.first; in[st.second.transIndex] = (uint16_t)iost.first; out[st.second.transIndex] = (uint16_t)ost.first; }
    for (size_t ti = 0; ti < n; ++ti) flat.addTransition((uint32_t)s, dest[ti], in[ti], out[ti], eval.state[s].logTransWeight[ti]);   // same walk as EvaluatedMachine::init
  }
  flat.finish();
}

// One device machine per EvaluatedMachine object: the matrices built from the same `eval` (the `for seqPair` loops of
// target/boss.cpp:796,826) share it, new weights at the same address (an EM loop re-evaluating in place) are re-sent with
// mb_machine_set_weights, a different topology rebuilds it.  forgetEvaluated() drops the entry when `eval` dies.
inline std::map<const void *, std::shared_ptr<FlatMachine>> &flatRegistry() { static std::map<const void *, std::shared_ptr<FlatMachine>> r; return r; }
inline void buildCharTable(const std::vector<std::string> &tok2sym, FlatMachine::CharTable &t) {
  for (int &x : t.tok) x = -1;
  t.usable = tok2sym.size() > 1;
  for (size_t k = 1; k < tok2sym.size(); ++k) {
    if (tok2sym[k].size() != 1) { t.usable = false; return; }
    t.tok[(unsigned char)tok2sym[k][0]] = (int)k;
  }
}
template <class SymbolT> void buildCharTable(const std::vector<SymbolT> &, FlatMachine::CharTable &t) { t.usable = false; }
// Tokenizer::tokenize (src/eval.h:29-41); a symbol the table does not know goes to the caller's tokenizer, which throws its own message
template <class TokenizerT>
std::vector<int> tokenizeWith(const FlatMachine::CharTable &t, const TokenizerT &tk, const std::vector<std::string> &seq) {
  if (!t.usable) return tk.tokenize(seq);
  std::vector<int> out(seq.size());
  for (size_t i = 0; i < seq.size(); ++i) {
    const int v = seq[i].size() == 1 ? t.tok[(unsigned char)seq[i][0]] : -1;
    if (v <= 0) return tk.tokenize(seq);
    out[i] = v;
  }
  return out;
}
template <class TokenizerT, class SymbolT>
std::vector<int> tokenizeWith(const FlatMachine::CharTable &, const TokenizerT &tk, const std::vector<SymbolT> &seq) { return tk.tokenize(seq); }

// does `flat` still describe `eval`?  (same walk as flattenEvaluated, nothing allocated); sameWeights tells whether only the weights moved
template <class EvalT>
bool sameTopology(const EvalT &eval, const FlatMachine &flat, bool &sameWeights) {
  sameWeights = true;
  if (flat.nStates != (int)eval.state.size() || flat.nInTok != (int)eval.inputTokenizer.tok2sym.size() - 1 || flat.nOutTok != (int)eval.outputTokenizer.tok2sym.size() - 1) return false;
  for (StateIndex s = 0; s < eval.state.size(); ++s) {
    const size_t n = eval.state[s].logTransWeight.size(), base = flat.transOffset[s];
    if (flat.transOffset[s + 1] - base != n) return false;
    size_t seen = 0;
    for (const auto &iost : eval.state[s].outgoing)
      for (const auto &ost : iost.second)
        for (const auto &st : ost.second) {
          const size_t e = base + st.second.transIndex;
          if (st.second.transIndex >= n || flat.dst[e] != (uint32_t)st.first || flat.inTok[e] != (uint16_t)iost.first || flat.outTok[e] != (uint16_t)ost.first) return false;
          ++seen;
        }
    if (seen != n) return false;
    for (size_t ti = 0; ti < n; ++ti) if (!(flat.logWeight[base + ti] == eval.state[s].logTransWeight[ti])) sameWeights = false;
  }
  return true;
}

template <class EvalT>
std::shared_ptr<FlatMachine> flatOf(const EvalT &eval) {
  std::shared_ptr<FlatMachine> &slot = flatRegistry()[(const void *)&eval];
  bool sameWeights = false;
  if (slot && sameTopology(eval, *slot, sameWeights)) {
    if (!sameWeights) {
      std::vector<double> lw;
      lw.reserve(slot->nTransitions());
      for (StateIndex s = 0; s < eval.state.size(); ++s) lw.insert(lw.end(), eval.state[s].logTransWeight.begin(), eval.state[s].logTransWeight.end());
      slot->setLogWeights(lw);
    }
    return slot;
  }
  slot = std::make_shared<FlatMachine>();
  flattenEvaluated(eval, *slot);
  buildCharTable(eval.inputTokenizer.tok2sym, slot->inChars);
  buildCharTable(eval.outputTokenizer.tok2sym, slot->outChars);
  return slot;
}
template <class EvalT>
void forgetEvaluated(const EvalT &eval) { flatRegistry().erase((const void *)&eval); }

// Envelope(const SeqPair&) (src/seqpair.cpp:104-110): the alignment's path envelope if the pair carries one, else full.
// alignment: a sequence of (inputSymbol, outputSymbol) columns, empty symbol = gap (MachinePath::AlignCol, src/machine.h:208).
template <class SeqPairT>
Envelope envelopeOf(const SeqPairT &sp) {
  Envelope e;
  if (!sp.alignment.empty()) {
    std::vector<Envelope::AlignCol> cols;
    for (const auto &c : sp.alignment) cols.push_back(Envelope::AlignCol(!c.first.empty(), !c.second.empty()));
    e.initPath(cols);
  } else e.initFull((long)sp.input.seq.size(), (long)sp.output.seq.size());
  return e;
}

// FNV-1a over the symbols of both tapes: the cheap proof that the SeqPair at a prefetched address is still the same pair
inline uint64_t fnvSeq(uint64_t h, const std::vector<std::string> &seq) {
  for (const std::string &sym : seq) {
    for (const char c : sym) { h ^= (unsigned char)c; h *= 1099511628211ull; }
    h ^= 0xFF; h *= 1099511628211ull;
  }
  return h;
}
template <class SymbolT> uint64_t fnvSeq(uint64_t h, const std::vector<SymbolT> &seq) {
  for (const SymbolT &sym : seq) { h ^= (uint64_t)std::hash<SymbolT>()(sym); h *= 1099511628211ull; }
  return h;
}
template <class SeqPairT> uint64_t fingerprintOf(const SeqPairT &sp) {
  uint64_t h = fnvSeq(14695981039346656037ull, sp.input.seq);
  h ^= 0xFE; h *= 1099511628211ull;
  return fnvSeq(h, sp.output.seq);
}

// What a matrix of the caller's types is constructed from: the device machine, the pair's tokens and Envelope(seqPair) -- and,
// if prefetch() ran for this SeqPair object and it has not changed since, its result (tokens are then taken from it)
struct Prepared {
  std::shared_ptr<FlatMachine> flat;
  std::vector<InputToken> in;
  std::vector<OutputToken> out;
  Envelope env;
  const PrefetchedPair *pre = nullptr;
};
template <class EvalT, class SeqPairT>
Prepared prepare(const EvalT &m, const SeqPairT &sp, int what) {
  Prepared p;
  p.flat = flatOf(m);
  p.env = envelopeOf(sp);
  const std::map<const void *, PrefetchedPair> *store = what == PrefetchLogLike ? &p.flat->prefetched.loglike : what == PrefetchViterbi ? &p.flat->prefetched.viterbi : nullptr;
  if (store && !store->empty()) {
    const auto it = store->find((const void *)&sp);
    if (it != store->end()) {
      const PrefetchedPair &pp = it->second;
      const bool full = p.env.isFull();
      if (pp.in.size() == sp.input.seq.size() && pp.out.size() == sp.output.seq.size() && pp.fingerprint == fingerprintOf(sp) &&
          (full ? pp.envStart.empty() : (pp.envStart == p.env.inStart && pp.envEnd == p.env.inEnd))) {
        p.in = pp.in; p.out = pp.out; p.pre = &pp;
        return p;
      }
    }
  }
  p.in = tokenizeWith(p.flat->inChars, m.inputTokenizer, sp.input.seq);
  p.out = tokenizeWith(p.flat->outChars, m.outputTokenizer, sp.output.seq);
  return p;
}

template <class EvalT, class SeqPairT, class Core>
class MatrixT : public Core {
protected:
  std::shared_ptr<FlatMachine> flatPtr;
  template <class... Extra>
  MatrixT(const EvalT &m, const SeqPairT &sp, const Prepared &p, Extra... extra)
      : Core(*p.flat, p.in, p.out, p.env, extra..., p.pre), flatPtr(p.flat), machine(m), seqPair(sp) {}
public:
  const EvalT &machine;       // the public members of the reference's DPMatrix (src/dpmatrix.h:125-131)
  const SeqPairT &seqPair;
  void writeJson(std::ostream &outs) const {
    this->writeJsonWith(outs, seqPair.input.name, seqPair.output.name, [this](std::ostream &o, StateIndex s) { o << machine.state[s].name; });
  }
  friend std::ostream &operator<<(std::ostream &out, const MatrixT &m) { m.writeJson(out); return out; }
};

// The reference's constructors (src/forward.h:19-27, backward.h:47-48, viterbi.h:13-14).  Like the reference's, the ones
// that take an Envelope IGNORE it and use Envelope(seqPair) (quirk Q1, src/dpmatrix.defs.h:16-17).
template <class EvalT, class SeqPairT>
class ForwardMatrixT : public MatrixT<EvalT, SeqPairT, ForwardCore> {
  typedef MatrixT<EvalT, SeqPairT, ForwardCore> Base;
public:
  ForwardMatrixT(const EvalT &m, const SeqPairT &sp) : Base(m, sp, prepare(m, sp, PrefetchLogLike), (StateIndex)0) {}
  template <class EnvT> ForwardMatrixT(const EvalT &m, const SeqPairT &sp, const EnvT &) : Base(m, sp, prepare(m, sp, PrefetchLogLike), (StateIndex)0) {}
  template <class EnvT> ForwardMatrixT(const EvalT &m, const SeqPairT &sp, const EnvT &, StateIndex startState) : Base(m, sp, prepare(m, sp, startState ? 0 : PrefetchLogLike), startState) {}
};
template <class EvalT, class SeqPairT>
class BackwardMatrixT : public MatrixT<EvalT, SeqPairT, BackwardCore> {
  typedef MatrixT<EvalT, SeqPairT, BackwardCore> Base;
public:
  BackwardMatrixT(const EvalT &m, const SeqPairT &sp) : Base(m, sp, prepare(m, sp, 0)) {}
  template <class EnvT> BackwardMatrixT(const EvalT &m, const SeqPairT &sp, const EnvT &) : Base(m, sp, prepare(m, sp, 0)) {}
  using BackwardCore::getCounts;
  template <class CountsT> static BackwardCore::BackTransVisitor transitionCounter(CountsT &counts) {      // src/backward.h:13-19
    return [&counts](StateIndex s, TransIndex ti, long, long, double postProb) { counts.count[s][ti] += postProb; };
  }
  // getCounts(forward, MachineCounts&), src/backward.cpp:58-60 (only for types with a `count` member: a visitor lvalue takes the overload above)
  template <class CountsT> auto getCounts(const ForwardCore &forward, CountsT &counts) const -> decltype(counts.count, void()) {
    BackwardCore::getCounts(forward, [&counts](StateIndex s, TransIndex ti, long, long, double postProb) { counts.count[s][ti] += postProb; });
  }
};
template <class EvalT, class SeqPairT>
class ViterbiMatrixT : public MatrixT<EvalT, SeqPairT, ViterbiCore> {
  typedef MatrixT<EvalT, SeqPairT, ViterbiCore> Base;
public:
  ViterbiMatrixT(const EvalT &m, const SeqPairT &sp) : Base(m, sp, prepare(m, sp, PrefetchViterbi)) {}
  template <class EnvT> ViterbiMatrixT(const EvalT &m, const SeqPairT &sp, const EnvT &) : Base(m, sp, prepare(m, sp, PrefetchViterbi)) {}
};
template <class EvalT, class SeqPairT>
class RollingOutputForwardMatrixT {        // MappedForwardMatrix<RollingOutputIndexMapper> (src/forward.h:29): logLike() only
  double ll;
public:
  RollingOutputForwardMatrixT(const EvalT &m, const SeqPairT &sp) {
    const Prepared p = prepare(m, sp, PrefetchLogLike);
    if (p.pre) { ll = p.pre->ll; return; }
    const TokSeqPair tsp{p.in, p.out};
    ll = RollingOutputForwardMatrix(*p.flat, tsp, &p.env).logLike();      // quirk Q1: Envelope(seqPair)
  }
  double logLike() const { return ll; }
};

// prefetch(eval, data.seqPairs, what): ONE batched device call over the pairs a caller is about to loop over; the
// ViterbiMatrix / ForwardMatrix / RollingOutputForwardMatrix objects it then constructs from the SAME SeqPair objects (the
// `for (const auto& seqPair : data.seqPairs)` loops of target/boss.cpp:796,826 bind references to the list's elements) take
// their score, path or log-likelihood from it instead of running one device call each.  what: PrefetchLogLike (rolling
// Forward), PrefetchViterbi (score + path), or both.  Pairs the machine cannot tokenise are skipped, as the loops skip them
// (`eval.canTokenize (seqPair)`).  New weights or another prefetch for the same EvaluatedMachine drop the old results.
template <class EvalT, class SeqPairsT>
void prefetch(const EvalT &eval, const SeqPairsT &seqPairs, int what) {
  const std::shared_ptr<FlatMachine> f = flatOf(eval);
  std::list<TokSeqPair> toks; std::list<Envelope> envs;
  std::vector<const TokSeqPair *> ps; std::vector<const Envelope *> es; std::vector<const void *> keys; std::vector<uint64_t> fps;
  for (const auto &sp : seqPairs) {
    // (the table tokenizer says no exactly when eval.canTokenize (sp) does -- two std::map look-ups per symbol, src/eval.h:23-28 --
    //  and then asks the caller's tokenizer, which throws)
    try { toks.push_back(TokSeqPair{tokenizeWith(f->inChars, eval.inputTokenizer, sp.input.seq), tokenizeWith(f->outChars, eval.outputTokenizer, sp.output.seq)}); }
    catch (const std::exception &) { continue; }
    envs.push_back(envelopeOf(sp));
    bool usable = envs.back().fits(toks.back()) && envs.back().connected();
    for (const int t : toks.back().input) usable = usable && t > 0;       // (an empty symbol tokenises to epsilon: the device rejects it)
    for (const int t : toks.back().output) usable = usable && t > 0;
    if (!usable) { toks.pop_back(); envs.pop_back(); continue; }          // the matrix constructor will throw for this one, as it would without a prefetch
    ps.push_back(&toks.back()); es.push_back(&envs.back()); keys.push_back((const void *)&sp); fps.push_back(fingerprintOf(sp));
  }
  prefetchPairs(*f, ps, es, keys, what, &fps);
}

// MachineCounts with the reference's whole surface (src/counts.h:11-25).  A SeqPairList goes to the device as ONE batch.
// PolicyT names the caller's weight algebra and string helper for the three members that never touch the DP
// (src/counts.cpp:73-106): static params(w, defs) / eval(w, defs) / deriv(w, defs, p) / asDouble(w) as in
// src/weight.h:83-90, and escaped_str(s) (src/util.h:100).  The glue header passes the reference's own WeightAlgebra.
struct NoAlgebraPolicy {};
template <class EvalT, class SeqPairT, class SeqPairListT, class PolicyT = NoAlgebraPolicy>
struct MachineCountsT : MachineCounts {
  MachineCountsT() {}
  MachineCountsT(const EvalT &m) { init(m); }
  MachineCountsT(const EvalT &m, const SeqPairT &sp) { init(m); (void)add(m, sp); }
  MachineCountsT(const EvalT &m, const SeqPairListT &l) { init(m); addList(m, l); }
  template <class EnvListT> MachineCountsT(const EvalT &m, const SeqPairListT &l, const EnvListT &) { init(m); addList(m, l); }   // envelopes: Envelope(seqPair) either way (Q1)
  void init(const EvalT &m) { MachineCounts::init(*flatOf(m)); }
  double add(const EvalT &m, const SeqPairT &sp) {
    const std::shared_ptr<FlatMachine> f = flatOf(m);
    const TokSeqPair tsp{tokenizeWith(f->inChars, m.inputTokenizer, sp.input.seq), tokenizeWith(f->outChars, m.outputTokenizer, sp.output.seq)};
    return MachineCounts::add(*f, {tsp}, {envelopeOf(sp)})[0];
  }
  template <class EnvT> double add(const EvalT &m, const SeqPairT &sp, const EnvT &) { return add(m, sp); }
  void addList(const EvalT &m, const SeqPairListT &l) {
    const std::shared_ptr<FlatMachine> f = flatOf(m);
    std::vector<TokSeqPair> pairs; std::vector<Envelope> envs;
    for (const auto &sp : l.seqPairs) {
      pairs.push_back(TokSeqPair{tokenizeWith(f->inChars, m.inputTokenizer, sp.input.seq), tokenizeWith(f->outChars, m.outputTokenizer, sp.output.seq)});
      envs.push_back(envelopeOf(sp));
    }
    (void)MachineCounts::add(*f, pairs, envs);
  }
  MachineCountsT &operator+=(const MachineCountsT &o) { MachineCounts::operator+=(o); return *this; }

  // writeJson (src/counts.cpp:73-78): one bracketed row per state, numbers at the stream default (6 significant digits)
  void writeJson(std::ostream &outs) const {
    outs << "[";
    for (size_t s = 0; s < count.size(); ++s) {
      std::ostringstream row;
      for (size_t t = 0; t < count[s].size(); ++t) row << (t ? "," : "") << count[s][t];
      outs << (s ? ",\n " : "") << "[" << row.str() << "]";
    }
    outs << "]" << std::endl;
  }
  // paramCounts (src/counts.cpp:89-106): expectation of d(logLike)/d(logParam) = sum over transitions of
  // count * (dw/dp) * p / w, through the caller's symbolic weights
  template <class MachineT, class ParamAssignT>
  std::map<std::string, double> paramCounts(const MachineT &machine, const ParamAssignT &prob) const {
    typedef typename std::decay<decltype(prob.defs)>::type ParamDefsT;
    std::map<std::string, double> paramCount;
    if (count.size() != machine.state.size()) throw std::runtime_error("Number of states mismatch");
    for (size_t s = 0; s < count.size(); ++s) {
      if (count[s].size() != machine.state[s].trans.size()) throw std::runtime_error("State size mismatch");
      auto transIter = machine.state[s].trans.begin();
      for (const double c : count[s]) {
        const auto &trans = *(transIter++);
        const auto transParams = PolicyT::params(trans.weight, ParamDefsT());
        const double w = PolicyT::eval(trans.weight, prob.defs);
        for (const auto &p : transParams) {
          const auto deriv = PolicyT::deriv(trans.weight, ParamDefsT(), p);
          paramCount[p] += c * PolicyT::eval(deriv, prob.defs) * PolicyT::asDouble(prob.defs.at(p)) / w;
        }
      }
    }
    return paramCount;
  }
  // writeParamCountsJson (src/counts.cpp:80-87)
  template <class MachineT, class ParamAssignT>
  void writeParamCountsJson(std::ostream &outs, const MachineT &machine, const ParamAssignT &prob) const {
    const std::map<std::string, double> pc = paramCounts(machine, prob);
    outs << "{";
    size_t n = 0;
    for (const auto &name_count : pc) outs << (n++ ? "," : "") << "\"" << PolicyT::escaped_str(name_count.first) << "\":" << name_count.second;
    outs << "}";
  }
};

}  // namespace MachineBossHIP

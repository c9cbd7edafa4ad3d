// boss.cpp / testforward.cpp / machine.cpp(downsample)-style caller code, written against the REFERENCE's class names and
// signatures only (hipdp.h binds them to the HIP engine; mock_reference.h stands in for the reference's eval.h / seqpair.h /
// machine.h).  Reads a machine and sequence pairs from a text file written by tests/test_cxx_glue.py, prints what the
// reference's own callers would read; the Python test checks every line against the CPU oracle.
#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>
#include <random>
#include <sstream>

#ifdef MB_GLUE_REAL      // parsed against the reference's own headers with the glue applied (apply_glue.py overlay)
#include "backward.h"
#include "viterbi.h"
#include "counts.h"
#include "api.h"
#else
#include "mock_reference.h"
#include "api_bodies.h"
#endif

using namespace MachineBoss;
using namespace std;

static string sym(const string &s) { return s == "-" ? string() : s; }

static void printPath(const char *tag, const MachinePath &p) {
  cout << tag;
  for (const auto &t : p.trans) cout << " " << t.dest << "," << (t.in.empty() ? "-" : t.in) << "," << (t.out.empty() ? "-" : t.out) << "," << setprecision(17) << t.weight;
  cout << endl;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
#ifdef MB_GLUE_REAL
  Machine machine = MachineLoader::fromFile(argv[1]);
  SeqPairList data = JsonLoader<SeqPairList>::fromFile(argv[2]);
  const size_t S = machine.nStates();
  ifstream f(argv[3]);
#else
  ifstream f(argv[1]);
  Machine machine;
  size_t S; f >> S;
  machine.state.resize(S);
  for (size_t s = 0; s < S; ++s) {
    size_t n; f >> machine.state[s].name.id >> n;
    for (size_t k = 0; k < n; ++k) { MachineTransition t; string in, out; f >> t.dest >> in >> out >> t.weight; t.in = sym(in); t.out = sym(out); machine.state[s].trans.push_back(t); }
  }
  SeqPairList data;
  size_t nPairs; f >> nPairs;
  for (size_t p = 0; p < nPairs; ++p) {
    SeqPair sp; size_t n;
    f >> sp.input.name >> sp.output.name >> n; sp.input.seq.resize(n); for (auto &x : sp.input.seq) f >> x;
    f >> n; sp.output.seq.resize(n); for (auto &x : sp.output.seq) f >> x;
    data.seqPairs.push_back(sp);
  }
#endif
  unsigned seed; f >> seed;
  const Params params;
  try {
    const EvaluatedMachine eval(machine, params);
    const SeqPair &seqPair = data.seqPairs.front();

    // t/src/testforward.cpp, testbackward.cpp
    const ForwardMatrix forward(eval, seqPair);
    cout << "FWDJSON_BEGIN" << endl; forward.writeJson(cout); cout << "FWDJSON_END" << endl;
    const BackwardMatrix backward(eval, seqPair);
    cout << "loglike " << setprecision(17) << forward.logLike() << " " << backward.logLike() << " dims " << forward.inLen << " " << forward.outLen << " " << forward.nStates
         << " tok " << forward.input.size() << " " << forward.output.size() << " outside " << forward.cell(forward.inLen + 1, 0, 0) << endl;

    // target/boss.cpp:796-805 (--loglike), :826-847 (--viterbi / --align)
    for (const auto &sp : data.seqPairs) {
      RollingOutputForwardMatrix roll(eval, sp);
      ViterbiMatrix viterbi(eval, sp);
      cout << "pair " << setprecision(17) << roll.logLike() << " " << viterbi.logLike() << endl;
      if (viterbi.logLike() > -numeric_limits<double>::infinity()) printPath("align", viterbi.path(machine));
    }

    // the same two loops behind the ONE added line of INTEGRATION.md section 2b: one batched device call, then the unchanged loop
    MachineBossHIP::prefetch(eval, data.seqPairs, MachineBossHIP::PrefetchLogLike | MachineBossHIP::PrefetchViterbi);
    for (const auto &sp : data.seqPairs) {
      RollingOutputForwardMatrix roll(eval, sp);
      ViterbiMatrix viterbi(eval, sp);
      cout << "ppair " << setprecision(17) << roll.logLike() << " " << viterbi.logLike() << endl;
      if (viterbi.logLike() > -numeric_limits<double>::infinity()) printPath("palign", viterbi.path(machine));
    }

    // target/boss.cpp:813, src/api.cpp:48-58 (--counts)
    MachineCounts counts(eval, data);
    cout << "counts " << setprecision(17) << counts.loglike;
    for (const auto &row : counts.count) for (double c : row) cout << " " << c;
    cout << endl;
    cout << "COUNTSJSON_BEGIN" << endl; counts.writeJson(cout); cout << "COUNTSJSON_END" << endl;      // t/src/testcounts.cpp:16
    cout << "paramcounts "; counts.writeParamCountsJson(cout, machine, ParamAssign()); cout << endl;    // target/boss.cpp:814
#ifdef MB_GLUE_REAL
    {   // src/fitter.cpp:30-33 (Baum-Welch iteration), target/boss.cpp:811-816, src/counts.cpp:89-106
      const list<Envelope> envelopes;
      const MachineCounts fitCounts(eval, data, envelopes);
      const Constraints constraints;
      const MachineObjective objective(machine, fitCounts, constraints, Params());
      const Params optParams = objective.optimize(params);
      const map<string, double> pc = fitCounts.paramCounts(machine, ParamAssign(optParams));
      fitCounts.writeParamCountsJson(cout, machine, params);
      MachineCounts sum(eval); sum += fitCounts;
      BackwardMatrix::BackTransVisitor tv = BackwardMatrix::transitionCounter(sum);
      backward.getCounts(forward, tv);
      cout << pc.size() << sum.add(eval, seqPair) << sum.add(eval, seqPair, Envelope(seqPair)) << endl;
    }
#endif
    MachineCounts one = forwardBackwardCounts(machine, params, seqPair), viaVisitor(eval);
    backward.getCounts(forward, viaVisitor);                                     // src/counts.cpp:57-64 spelled out
    double d = 0; for (size_t s = 0; s < one.count.size(); ++s) for (size_t t = 0; t < one.count[s].size(); ++t) d = max(d, fabs(one.count[s][t] - viaVisitor.count[s][t]));
    cout << "counts_visitor_vs_device " << d << " api " << forwardLogLike(machine, params, seqPair) << " " << viterbiLogLike(machine, params, seqPair) << endl;
    const bool reachable = forward.logLike() > -numeric_limits<double>::infinity();      // boss.cpp checks the same before tracing (target/boss.cpp:831)
    if (reachable) printPath("apialign", viterbiAlign(machine, params, seqPair));
    else { try { (void)viterbiAlign(machine, params, seqPair); cout << "noalign none" << endl; } catch (const runtime_error &e) { cout << "noalign " << e.what() << endl; } }
    if (reachable) {

    // ForwardMatrix::samplePath (src/forward.cpp:17-23), stochasticDownsample's loop (src/machine.cpp:2107-2122)
    mt19937 rng(seed);
    for (int k = 0; k < 3; ++k) printPath("sample", forward.samplePath(machine, rng));
    {
      mt19937 rng2(seed + 1);
      MachinePath mp;
      DPMatrix<IdentityIndexMapper>::TraceTerminator neverStopTrace = [&](Envelope::InputIndex, Envelope::OutputIndex, StateIndex s, EvaluatedMachineState::TransIndex ti) {
        mp.trans.push_front(machine.state[s].getTransition(ti)); return false; };
      ForwardMatrix::TransSelector selectRandomTrans = forward.randomTransSelector(rng2);
      forward.traceBack(machine, forward.inLen, forward.outLen, machine.endState(), neverStopTrace, selectRandomTrans);
      printPath("sample2", mp);
    }

    // quirk Q2 overloads
    // (on a random machine these odd walks may start from a -inf cell or run into a cell without candidates -- an Assert /
    //  undefined behaviour in the reference, a runtime_error here; the oracle must fail on exactly the same ones)
    auto attempt = [&](const char *tag, function<MachinePath()> walk) {
      try { printPath(tag, walk()); } catch (const runtime_error &e) { cout << tag << "_error " << e.what() << endl; }
    };
    attempt("tb_state", [&]() { return forward.traceBack(machine, (StateIndex)(S - 1)); });
    attempt("tf_quirk", [&]() { return backward.traceForward(machine); });
    attempt("tf_pos", [&]() { return backward.traceForward(machine, 0, 0, (StateIndex)(S - 1)); });
    attempt("tracefrom3", [&]() { return backward.traceFrom(machine, forward, 0, 0, (StateIndex)(S - 1)); });

    BackwardMatrix::PostTransQueue fullQueue = backward.postTransQueue(forward);
    cout << "queue " << fullQueue.size() << " top " << setprecision(17) << (fullQueue.empty() ? 0.0 : fullQueue.top().weight) << endl;
    }
    // Machine::downsample (src/machine.cpp:2036-2076), for machines it accepts (acyclic, topologically sorted): the labels are
    // stripped, the matrices are those of the EMPTY sequence pair, and the posterior queue is drained through traceFrom
    bool dag = true;
    for (size_t s = 0; s < S; ++s) for (const auto &t : machine.state[s].trans) dag = dag && t.dest > s;
    if (dag) {
      Machine null(machine);
      vector<vector<bool>> transAllowed;
      for (auto &ms : null.state) { for (auto &mt : ms.trans) mt.in = mt.out = string(); transAllowed.push_back(vector<bool>(ms.trans.size())); }
      const SeqPair emptySeqPair;
      const EvaluatedMachine evalNull(null, params);
      const ForwardMatrix fwd(evalNull, emptySeqPair);
      const BackwardMatrix back(evalNull, emptySeqPair);
      size_t nTrans = 0;
      DPMatrix<IdentityIndexMapper>::TraceTerminator stopTrace = [&](Envelope::InputIndex, Envelope::OutputIndex, StateIndex s, EvaluatedMachineState::TransIndex ti) {
        if (transAllowed[s][ti]) return true;
        transAllowed[s][ti] = true; ++nTrans; return false; };
      BackwardMatrix::PostTransQueue queue = back.postTransQueue(fwd);
      cout << "nullqueue " << queue.size() << " " << setprecision(17) << fwd.logLike() << endl;
      const size_t nTransTarget = (size_t)(null.state.size() ? 0.6 * queue.size() : 0);
      while (!queue.empty() && (nTrans == 0 || nTrans < nTransTarget)) {
        const BackwardMatrix::PostTrans pt = queue.top();
        queue.pop();
        back.traceFrom(null, fwd, pt.inPos, pt.outPos, pt.src, pt.transIndex, stopTrace);
        cout << "pop " << pt.inPos << " " << pt.outPos << " " << pt.src << " " << pt.transIndex << " " << setprecision(17) << pt.weight << " allowed";
        for (auto &row : transAllowed) for (bool b : row) cout << " " << (b ? 1 : 0);
        cout << endl;
      }
    }
    // errors surface as runtime_error with the reference's messages
    try { SeqPair bad = seqPair; bad.input.seq.push_back("?"); ForwardMatrix oops(eval, bad); cout << "error none" << endl; }
    catch (const runtime_error &e) { cout << "error " << e.what() << endl; }
  } catch (const exception &e) { cout << "EXCEPTION " << e.what() << endl; return 1; }
  cout << "GLUE OK" << endl;
  return 0;
}

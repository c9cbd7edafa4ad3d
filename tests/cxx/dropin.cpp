// The reference's own call sites, run through the drop-in classes: the `--loglike` loop of target/boss.cpp:796-800 and the
// `--viterbi / --align` loop of target/boss.cpp:826-833 exactly as they are written there (one matrix object per pair), with
// and without the ONE line INTEGRATION.md section 2b adds in front of each (MachineBossHIP::prefetch), next to the batch C-ABI
// no reference caller uses.  Written against the reference's class names only (hipdp.h binds them; mock_reference.h stands in
// for eval.h / seqpair.h / machine.h, as in test_glue.cpp).
//
//   dropin <case.txt> time  <reps>   -> one JSON line per variant: seconds of each loop (best of reps), fp64 matrices fetched
//   dropin <case.txt> check          -> per pair: scores, paths, then cell() of every cell AFTER the lazy construction
//
// bench.py (extra.dropin) runs the first form on BASELINE configs 2 and 4; tests/test_dropin.py checks the second against the oracle.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

#include "mock_reference.h"
#include "api_bodies.h"

using namespace MachineBoss;
using namespace std;

static string sym(const string &s) { return s == "-" ? string() : s; }
static double now() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

struct Loaded { Machine machine; SeqPairList data; };

static void load(const char *path, Loaded &L) {
  ifstream f(path);
  if (!f) throw runtime_error(string("cannot open ") + path);
  size_t S; f >> S;
  L.machine.state.resize(S);
  for (size_t s = 0; s < S; ++s) {
    size_t n; f >> L.machine.state[s].name.id >> n;
    for (size_t k = 0; k < n; ++k) { MachineTransition t; string in, out; f >> t.dest >> in >> out >> t.weight; t.in = sym(in); t.out = sym(out); L.machine.state[s].trans.push_back(t); }
  }
  size_t nPairs; f >> nPairs;
  for (size_t p = 0; p < nPairs; ++p) {
    SeqPair sp; size_t n;
    f >> sp.input.name >> sp.output.name >> n; sp.input.seq.resize(n); for (auto &x : sp.input.seq) f >> x;
    f >> n; sp.output.seq.resize(n); for (auto &x : sp.output.seq) f >> x;
    L.data.seqPairs.push_back(sp);
  }
}

struct LoopResult { double llSum = 0, vitSum = 0; size_t pathLen = 0, aligned = 0; double prefetchLL = 0, prefetchVit = 0, pathObjects = 0; };   // seconds inside the added line / inside viterbi.path(machine)

// target/boss.cpp:793-808, output text left out (the numbers it would print are summed instead)
static void loglikeLoop(const Machine &machine, const Params &params, const SeqPairList &data, bool withPrefetch, LoopResult &r, ostream *out) {
  const EvaluatedMachine eval(machine, params);
  const double t0 = now();
  if (withPrefetch) MachineBossHIP::prefetch(eval, data.seqPairs, MachineBossHIP::PrefetchLogLike);     // <- the added line
  r.prefetchLL = now() - t0;
  for (const auto &seqPair : data.seqPairs) {
    double fwdLogLike = -numeric_limits<double>::infinity();
    if (eval.canTokenize(seqPair)) {
      const RollingOutputForwardMatrix forward(eval, seqPair);
      fwdLogLike = forward.logLike();
    }
    r.llSum += fwdLogLike;
    if (out) *out << "loglike " << setprecision(17) << fwdLogLike << endl;
  }
}

// target/boss.cpp:819-847
static void alignLoop(const Machine &machine, const Params &params, const SeqPairList &data, bool withPrefetch, LoopResult &r, ostream *out) {
  const EvaluatedMachine eval(machine, params);
  const double t0 = now();
  if (withPrefetch) MachineBossHIP::prefetch(eval, data.seqPairs, MachineBossHIP::PrefetchViterbi);     // <- the added line
  r.prefetchVit = now() - t0;
  for (const auto &seqPair : data.seqPairs) {
    double vitLogLike = -numeric_limits<double>::infinity();
    if (eval.canTokenize(seqPair)) {
      const ViterbiMatrix viterbi(eval, seqPair);
      vitLogLike = viterbi.logLike();
      if (vitLogLike > -numeric_limits<double>::infinity()) {
        const double tp = now();
        const MachinePath path = viterbi.path(machine);
        r.pathObjects += now() - tp;
        r.pathLen += path.trans.size(); ++r.aligned;
        if (out) {
          *out << "align";
          for (const auto &t : path.trans) *out << " " << t.dest << "," << (t.in.empty() ? "-" : t.in) << "," << (t.out.empty() ? "-" : t.out);
          *out << endl;
        }
      }
    }
    r.vitSum += vitLogLike;
    if (out) *out << "viterbi " << setprecision(17) << vitLogLike << endl;
  }
}

// the batch C-ABI on the same data, from the same SeqPairList (tokenising included, as any batch caller must): what bench.py's
// other figures go through (there the tokens are resident before the clock starts)
static void batchCalls(const Machine &machine, const Params &params, const SeqPairList &data, LoopResult &r, double &tLL, double &tVit) {
  const EvaluatedMachine eval(machine, params);
  for (int which = 0; which < 2; ++which) {
    const double t0 = now();
    const std::shared_ptr<MachineBossHIP::FlatMachine> f = MachineBossHIP::flatOf(eval);
    list<MachineBossHIP::TokSeqPair> toks;
    vector<const MachineBossHIP::TokSeqPair *> ps;
    for (const auto &sp : data.seqPairs)
      if (eval.canTokenize(sp)) {
        toks.push_back(MachineBossHIP::TokSeqPair{MachineBossHIP::tokenizeWith(f->inChars, eval.inputTokenizer, sp.input.seq), MachineBossHIP::tokenizeWith(f->outChars, eval.outputTokenizer, sp.output.seq)});
        ps.push_back(&toks.back());
      }
    MachineBossHIP::DeviceBatch b(*f, ps, {});
    if (which == 0) { for (double l : b.forward(MB_ROLLING)) r.llSum += l; tLL = now() - t0; }
    else {
      vector<double> ll; vector<int64_t> off; vector<uint32_t> edges; b.viterbi(ll, off, edges);
      for (double l : ll) r.vitSum += l;
      r.pathLen += edges.size(); r.aligned += ll.size();
      tVit = now() - t0;
    }
  }
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: dropin <case.txt> time <reps> | check\n"); return 2; }
  try {
    Loaded L;
    load(argv[1], L);
    const Params params;
    const string mode = argv[2];
    double cells = 0;
    for (const auto &sp : L.data.seqPairs) cells += (double)(sp.input.seq.size() + 1) * (sp.output.seq.size() + 1) * L.machine.nStates();
    if (mode == "time") {
      const int reps = argc > 3 ? atoi(argv[3]) : 3;
      {   // warm-up: run-time compilation (or the code-object cache), pools
        SeqPairList one; one.seqPairs.push_back(L.data.seqPairs.front());
        LoopResult w; loglikeLoop(L.machine, params, one, false, w, nullptr); alignLoop(L.machine, params, one, false, w, nullptr);
        double a, b; batchCalls(L.machine, params, L.data, w, a, b);
      }
      for (int variant = 0; variant < 3; ++variant) {
        double bestLL = 1e30, bestVit = 1e30; LoopResult r, best; long fills0 = MachineBossHIP::matrixFills();
        for (int k = 0; k < reps; ++k) {
          r = LoopResult();
          double tLL, tVit;
          if (variant < 2) {
            double t0 = now(); loglikeLoop(L.machine, params, L.data, variant == 1, r, nullptr); tLL = now() - t0;
            t0 = now(); alignLoop(L.machine, params, L.data, variant == 1, r, nullptr); tVit = now() - t0;
          } else batchCalls(L.machine, params, L.data, r, tLL, tVit);
          if (tLL + tVit < bestLL + bestVit) best = r;
          bestLL = min(bestLL, tLL); bestVit = min(bestVit, tVit);
        }
        printf("{\"variant\": \"%s\", \"pairs\": %zu, \"cells\": %.0f, \"loglike_s\": %.6f, \"align_s\": %.6f, \"loglike_gcells\": %.3f, \"align_gcells\": %.3f, "
               "\"matrix_fills\": %ld, \"ll_sum\": %.10f, \"vit_sum\": %.10f, \"path_transitions\": %zu, \"aligned\": %zu, "
               "\"prefetch_loglike_s\": %.6f, \"prefetch_align_s\": %.6f, \"path_objects_s\": %.6f}\n",
               variant == 0 ? "unchanged_loop" : variant == 1 ? "unchanged_loop_prefetch" : "batch_c_abi", L.data.seqPairs.size(), cells, bestLL, bestVit,
               cells / bestLL * 1e-9, cells / bestVit * 1e-9, MachineBossHIP::matrixFills() - fills0, r.llSum, r.vitSum, r.pathLen, r.aligned,
               best.prefetchLL, best.prefetchVit, best.pathObjects);
        fflush(stdout);
      }
    } else if (mode == "check") {
      LoopResult r;
      cout << "LOOP" << endl;
      loglikeLoop(L.machine, params, L.data, false, r, &cout); alignLoop(L.machine, params, L.data, false, r, &cout);
      cout << "fills_after_loops " << MachineBossHIP::matrixFills() << endl;
      cout << "PREFETCH" << endl;
      loglikeLoop(L.machine, params, L.data, true, r, &cout); alignLoop(L.machine, params, L.data, true, r, &cout);
      cout << "fills_after_prefetch_loops " << MachineBossHIP::matrixFills() << endl;
      // cell() after a lazy construction, src/api.cpp's wrappers, and a matrix constructed from a prefetched pair
      const EvaluatedMachine eval(L.machine, params);
      MachineBossHIP::prefetch(eval, L.data.seqPairs, MachineBossHIP::PrefetchViterbi | MachineBossHIP::PrefetchLogLike);
      size_t k = 0;
      for (const auto &seqPair : L.data.seqPairs) {
        if (!eval.canTokenize(seqPair)) continue;
        const ViterbiMatrix viterbi(eval, seqPair);
        const ForwardMatrix forward(eval, seqPair);
        const long before = MachineBossHIP::matrixFills();
        cout << "lazy " << k << " " << setprecision(17) << viterbi.logLike() << " " << forward.logLike() << " fetched " << viterbi.matrixFetched() << forward.matrixFetched() << endl;
        cout << "vcells " << k;
        for (long o = 0; o <= viterbi.outLen; ++o) for (long i = 0; i <= viterbi.inLen; ++i) for (StateIndex s = 0; s < viterbi.nStates; ++s) cout << " " << setprecision(17) << viterbi.cell(i, o, s);
        cout << endl << "fcells " << k;
        for (long o = 0; o <= forward.outLen; ++o) for (long i = 0; i <= forward.inLen; ++i) for (StateIndex s = 0; s < forward.nStates; ++s) cout << " " << setprecision(17) << forward.cell(i, o, s);
        cout << endl << "fills_for_pair " << MachineBossHIP::matrixFills() - before << " endcells " << setprecision(17) << viterbi.endCell() << " " << forward.endCell() << endl;
        if (viterbi.logLike() > -numeric_limits<double>::infinity()) {      // the host walker over the fetched matrix agrees with the device's path
          const MachinePath a = viterbi.path(L.machine), b = viterbi.traceBack(L.machine);
          bool same = a.trans.size() == b.trans.size();
          auto ia = a.trans.begin(); auto ib = b.trans.begin();
          for (; same && ia != a.trans.end(); ++ia, ++ib) same = ia->dest == ib->dest && ia->in == ib->in && ia->out == ib->out && ia->weight == ib->weight;
          cout << "walker_agrees " << (same ? 1 : 0) << endl;
        }
        ++k;
      }
      const SeqPair &first = L.data.seqPairs.front();
      cout << "api " << setprecision(17) << forwardLogLike(L.machine, params, first) << " " << viterbiLogLike(L.machine, params, first) << " " << viterbiAlign(L.machine, params, first).trans.size() << endl;
    } else { fprintf(stderr, "unknown mode\n"); return 2; }
  } catch (const exception &e) { cout << "EXCEPTION " << e.what() << endl; return 1; }
  cout << "DROPIN OK" << endl;
  return 0;
}

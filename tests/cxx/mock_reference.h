// mock_reference.h -- TEST SCAFFOLDING: the SHAPES of the reference's host types that the DP classes touch
// (src/machine.h, src/eval.h, src/seqpair.h of Machine Boss), written from their public interface so that the glue of
// INTEGRATION.md section 2 can be compiled and exercised here without the reference's sources (which need GSL / Boost /
// nlohmann-json and do not travel to the GPU box).  Member names, nesting and semantics follow the reference; weights are
// plain numbers (the reference evaluates WeightExpr trees -- irrelevant to the DP path).
#pragma once
#include <cmath>
#include <list>
#include <map>
#include <ostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace MachineBoss {

typedef std::string InputSymbol;
typedef std::string OutputSymbol;
typedef unsigned long long StateIndex;
typedef std::map<std::string, double> Params;

// weights are plain numbers here: the mock algebra has no parameters (src/weight.h:83-90, src/params.h:26-30, src/util.h:100)
typedef double WeightExpr;
typedef std::map<std::string, WeightExpr> ParamDefs;
struct ParamAssign { ParamDefs defs; };
struct WeightAlgebra {
  static std::set<std::string> params(const WeightExpr &, const ParamDefs &) { return std::set<std::string>(); }
  static double eval(const WeightExpr &w, const ParamDefs &) { return w; }
  static WeightExpr deriv(const WeightExpr &, const ParamDefs &, const std::string &) { return 0; }
  static double asDouble(const WeightExpr &w) { return w; }
};
inline std::string escaped_str(const std::string &s) { return s; }

struct StateName {                       // nlohmann::json in the reference: streams as a JSON value
  std::string id;
  bool is_null() const { return id.empty(); }
};
inline std::ostream &operator<<(std::ostream &o, const StateName &n) { return o << "\"" << n.id << "\""; }

struct MachineTransition {               // src/machine.h: in, out, dest, weight
  InputSymbol in; OutputSymbol out; StateIndex dest; double weight;
  bool inputEmpty() const { return in.empty(); }
  bool outputEmpty() const { return out.empty(); }
};
typedef std::list<MachineTransition> TransList;

struct MachineState {
  StateName name;
  TransList trans;
  const MachineTransition &getTransition(size_t n) const { auto it = trans.begin(); std::advance(it, (long)n); return *it; }
};

struct MachinePath {                     // src/machine.h:207-220
  typedef std::pair<InputSymbol, OutputSymbol> AlignCol;
  typedef std::list<AlignCol> AlignPath;
  TransList trans;
  MachinePath() {}
  MachinePath(const MachineTransition &t) : trans(1, t) {}
  void clear() { trans.clear(); }
  MachinePath concatenate(const MachinePath &m) const { MachinePath r(*this); r.trans.insert(r.trans.end(), m.trans.begin(), m.trans.end()); return r; }
};

struct Machine {
  std::vector<MachineState> state;
  StateIndex nStates() const { return state.size(); }
  StateIndex endState() const { return state.size() - 1; }
  std::vector<InputSymbol> inputAlphabet() const { std::set<std::string> a; for (auto &s : state) for (auto &t : s.trans) if (!t.in.empty()) a.insert(t.in); return std::vector<std::string>(a.begin(), a.end()); }
  std::vector<OutputSymbol> outputAlphabet() const { std::set<std::string> a; for (auto &s : state) for (auto &t : s.trans) if (!t.out.empty()) a.insert(t.out); return std::vector<std::string>(a.begin(), a.end()); }
};

template <typename Symbol, typename Token>
struct Tokenizer {                       // src/eval.h:11-50
  std::vector<Symbol> tok2sym;
  std::map<Symbol, Token> sym2tok;
  Tokenizer() {}
  Tokenizer(const std::vector<Symbol> &symbols) {
    tok2sym.push_back(Symbol());
    tok2sym.insert(tok2sym.end(), symbols.begin(), symbols.end());
    for (Token tok = 0; tok < (Token)tok2sym.size(); ++tok) sym2tok[tok2sym[tok]] = tok;
  }
  static Token emptyToken() { return 0; }
  bool canTokenize(const std::vector<Symbol> &symSeq) const {          // src/eval.h:23-28
    for (const auto &sym : symSeq) if (!sym2tok.count(sym)) return false;
    return true;
  }
  std::vector<Token> tokenize(const std::vector<Symbol> &symSeq) const {
    std::vector<Token> tokSeq;
    for (const auto &sym : symSeq) {
      if (!sym2tok.count(sym)) throw std::runtime_error("Can't tokenize symbol " + sym + " using this alphabet");
      tokSeq.push_back(sym2tok.at(sym));
    }
    return tokSeq;
  }
};
typedef int InputToken;
typedef int OutputToken;
typedef Tokenizer<InputSymbol, InputToken> InputTokenizer;
typedef Tokenizer<OutputSymbol, OutputToken> OutputTokenizer;
typedef double LogWeight;

struct EvaluatedMachineState {           // src/eval.h:59-75
  typedef size_t TransIndex;
  struct Trans { LogWeight logWeight; TransIndex transIndex; };
  typedef std::multimap<StateIndex, Trans> StateTransMap;
  typedef std::map<OutputToken, StateTransMap> OutStateTransMap;
  typedef std::map<InputToken, OutStateTransMap> InOutStateTransMap;
  StateName name;
  TransIndex nTransitions, transOffset;
  InOutStateTransMap incoming, outgoing;
  std::vector<LogWeight> logTransWeight;
};

struct EvaluatedMachine {                // src/eval.h:77-98, init as src/eval.cpp:40-70
  InputTokenizer inputTokenizer;
  OutputTokenizer outputTokenizer;
  std::vector<EvaluatedMachineState> state;
  EvaluatedMachineState::TransIndex nTransitions;
  EvaluatedMachine(const Machine &machine, const Params &)
      : inputTokenizer(machine.inputAlphabet()), outputTokenizer(machine.outputAlphabet()), state(machine.nStates()) {
    EvaluatedMachineState::TransIndex tiCum = 0;
    for (StateIndex s = 0; s < nStates(); ++s) {
      state[s].name = machine.state[s].name;
      EvaluatedMachineState::TransIndex ti = 0;
      for (const auto &trans : machine.state[s].trans) {
        const InputToken in = inputTokenizer.sym2tok.at(trans.in);
        const OutputToken out = outputTokenizer.sym2tok.at(trans.out);
        const LogWeight lw = std::log(trans.weight);
        state[s].outgoing[in][out].insert(EvaluatedMachineState::StateTransMap::value_type(trans.dest, EvaluatedMachineState::Trans{lw, ti}));
        state[trans.dest].incoming[in][out].insert(EvaluatedMachineState::StateTransMap::value_type(s, EvaluatedMachineState::Trans{lw, ti}));
        state[s].logTransWeight.push_back(lw);
        ++ti;
      }
      state[s].nTransitions = ti; state[s].transOffset = tiCum; tiCum += ti;
    }
    nTransitions = tiCum;
  }
  StateIndex nStates() const { return state.size(); }
  StateIndex startState() const { return 0; }
  StateIndex endState() const { return nStates() - 1; }
  template <class SeqPairT> bool canTokenize(const SeqPairT &sp) const {   // src/eval.cpp:218-220
    return inputTokenizer.canTokenize(sp.input.seq) && outputTokenizer.canTokenize(sp.output.seq);
  }
};

template <typename Symbol>
struct NamedSeq { std::string name; std::vector<Symbol> seq; };

struct SeqPair {                         // src/seqpair.h:56-73
  typedef MachinePath::AlignCol AlignCol;
  typedef MachinePath::AlignPath AlignPath;
  NamedSeq<InputSymbol> input;
  NamedSeq<OutputSymbol> output;
  AlignPath alignment;
};

struct Envelope {                        // src/seqpair.h:75-116 (the members the DP classes read)
  typedef long InputIndex;
  typedef long OutputIndex;
  InputIndex inLen; OutputIndex outLen;
  std::vector<InputIndex> inStart, inEnd;
  Envelope() : inLen(0), outLen(0) {}
  Envelope(const SeqPair &sp) : inLen((long)sp.input.seq.size()), outLen((long)sp.output.seq.size()), inStart(outLen + 1, 0), inEnd(outLen + 1, inLen + 1) {}
};

struct SeqPairList { std::list<SeqPair> seqPairs; };

}  // namespace MachineBoss

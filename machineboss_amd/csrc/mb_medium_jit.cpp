// mb_medium_jit.cpp -- run-time specialisation of the "lanes = states" tile kernel with hiprtc (see
// mb_medium_jit_src.h for the rationale).  The generator unrolls the compiled program of ONE machine into
// straight-line HIP; everything structural becomes a literal, and every candidate record gets a PLACEMENT:
//
//   REG     records that do not change along the sweep of a column -- token-independent ones (silent closure) and the
//           ones selected by the column's INPUT token only -- are loaded once into VGPRs before the step loop;
//   LDS     records selected by the output token (and whatever token-independent ones exceed the VGPR budget) are
//           copied into LDS next to the ring at the start of a tile;
//   GLOBAL  what fits neither (large match tables) is fetched per step with global_load_dwordx4 as before.
//
// With REG + LDS placement the step loop issues no vector-memory LOAD other than the halo supercell, so nothing in
// a step waits (through the in-order vmcnt counter of gfx9) for the previous step's global stores to be acknowledged.

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>

#include "mb_jit.h"
#include "mb_medium.h"
#include "mb_medium_jit_src.h"

namespace mb {

static int env_int_early(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}
static const int JIT_MAX_CANDS = std::max(2, env_int_early("MB_JIT_MAXCANDS", 12));   // candidates evaluated in one straight-line round body

static int env_int(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}
int medium_jit_max_cands() { return JIT_MAX_CANDS; }
// MB_JIT_DEBUG (kernels with switched-off loads: WRONG results by design) exists only in a library built with
// -DMB_EXPERIMENTS; a product build ignores the variable, so a stray environment cannot corrupt counts or likelihoods
static int jit_debug_bits() {
#ifdef MB_EXPERIMENTS
  const int v = env_int("MB_JIT_DEBUG", 0);
  if (v & 3) { static bool told = false; if (!told) { told = true; fprintf(stderr, "[mbhip] WARNING: MB_JIT_DEBUG=%d -- tile kernels skip loads, results are WRONG by design\n", v); } }
  return v;
#else
  return 0;
#endif
}

static size_t ring_doubles(const MedProgram &P, const MedGeom &geo) {   // even: the record image follows, 16-byte aligned
  const size_t perCol = geo.compact ? (size_t)P.Spad + (size_t)P.NS * medium_compact_len(P) : (size_t)P.NS * P.Spad;      // compact ring: one full vector + NS short ones
  return (perCol * (geo.C + 1) + 1) & ~(size_t)1;
}
static size_t ring_bytes(const MedProgram &P, const MedGeom &geo) { return ring_doubles(P, geo) * sizeof(double); }
static size_t tok_bytes(const MedProgram &P, const MedGeom &geo) { return 3 * 2ull * (size_t)(P.tokWindow + geo.C) * sizeof(int); }   // output tokens + the envelope rows (start, end) of the same window

// count programs: one Backward supercell per column + the loop-time count accumulators.  A flat program's all-transition table (one
// entry per transition + LPG dummies) exists only AFTER the step loop and is laid over the start of the LDS (ring, records, tokens,
// Backward supercells: all dead by then); the loop-time table must not lie under it, so it starts no earlier than that table ends
static size_t acc_offset(const MedProgram &P, const MedGeom &geo) {
  const size_t natural = ring_bytes(P, geo) + P.ldsImageIdx.size() * sizeof(MedRec) + tok_bytes(P, geo) + (size_t)geo.C * P.Spad * sizeof(double);
  return std::max(natural, ((size_t)P.accAllEntries * sizeof(double) + 15) & ~(size_t)15);
}
static size_t count_bytes(const MedProgram &P, const MedGeom &geo) {
  return P.counting ? ((size_t)geo.C * P.Spad + (size_t)P.accEntries) * sizeof(double) : 0;
}

static size_t halo_bytes(const MedProgram &P, const MedGeom &geo) { return (size_t)geo.haloSteps * P.dev.S * sizeof(double); }

// traceback-byte Viterbi: one byte vector per column (the step's codes, copied out 16 bytes per lane), + alignment slack
static int tb_lds_stride(const MedProgram &P) { return (P.Spad + 15) & ~15; }
static size_t tb_bytes(const MedProgram &P, const MedGeom &geo, int mode) { return mode == MED_MODE_TB ? (size_t)geo.C * tb_lds_stride(P) + 16 : 0; }

// (+ 128 bytes behind everything else: the per-wavefront step counters of the neighbour synchronisation, JNBSYNC)
static size_t lds_payload_bytes(const MedProgram &P, const MedGeom &geo, int mode) {
  if (P.counting) return (acc_offset(P, geo) + (size_t)P.accEntries * sizeof(double) + halo_bytes(P, geo) + 15) & ~(size_t)15;
  return (ring_bytes(P, geo) + P.ldsImageIdx.size() * sizeof(MedRec) + tok_bytes(P, geo) + count_bytes(P, geo) + halo_bytes(P, geo) + tb_bytes(P, geo, mode) + 15) & ~(size_t)15;
}
size_t medium_jit_lds_bytes(const MedProgram &P, const MedGeom &geo, int mode) { return lds_payload_bytes(P, geo, mode) + 128; }

// Traceback bytes: code = table << 6 | index of the candidate in its table's list, so a state may have at most 64
// candidates per table (any machine with a larger fan-in keeps the fp64 Viterbi matrix); a silent self-loop on state 0 --
// the one non-advancing transition the reference lets through -- is a candidate of its traceback but not of the fill.
bool medium_tb_eligible(const mb_machine *m, const MedProgram &P) {
  if (P.closure || P.backward || P.counting || P.hasSplits) return false;
  for (long long e = 0; e < m->nTrans; ++e)
    if (m->inTok[e] == 0 && m->outTok[e] == 0 && m->dst[e] <= m->src[e]) return false;
  for (const MedRoundInfo &ri : P.roundInfo) {
    int perT[4] = {0, 0, 0, 0};
    for (const MedSlotInfo &sl : ri.slots) if (++perT[sl.T] > 64) return false;
  }
  return true;
}

// Decide where every slot's records live (see the file header).  Deterministic in (program, geometry).
void medium_jit_plan(const mb_machine *m, MedProgram &P, const MedGeom &geoIn) {
  // One placement serves every kernel of the program -- all strip widths (medium_pick_geometry), with and without halo
  // supercells -- so it is made for the geometry the program was built with (the widest strips, the largest ring): a
  // re-plan after a spill, called with the narrower strips of a short batch, otherwise fills the LDS those strips leave
  // free and the wider kernels no longer fit (they then ran the ahead-of-time kernel: right, but slow).
  if (P.planC == 0) { P.planC = geoIn.C; P.planHalo = geoIn.haloSteps; P.planWaves = geoIn.waves; }
  MedGeom geo = geoIn;
  geo.C = std::max(geo.C, P.planC); geo.haloSteps = std::max(geo.haloSteps, P.planHalo); geo.waves = std::max(geo.waves, P.planWaves);
  if (P.regBudget < 0) {
    // VGPRs left for loop-invariant records: a wavefront of a W-wave workgroup may use 512 / ceil(W / 4) registers and
    // the specialised kernel needs ~90 of them for everything else; medium_jit_get lowers this if the compiler spills
    const int perWave = std::min(512 / ((geo.waves + 3) / 4), 256);
    P.regBudget = env_int("MB_JIT_REGBUDGET", std::max(0, perWave - 84));
  }
  P.ldsImageIdx.clear();
  P.tokWindow = 64;
  const int LPG = P.LPG;
  const long long ntokT[4] = {(long long)(m->nIn + 1) * (m->nOut + 1), m->nIn + 1, m->nOut + 1, 1};
  // (the traceback-byte vectors of MED_MODE_TB share the plan of the program's other kernels: reserved whenever they could be used)
  long long ldsFree = 160 * 1024 - 256 - (long long)ring_bytes(P, geo) - (long long)tok_bytes(P, geo) - (long long)count_bytes(P, geo) - (long long)halo_bytes(P, geo)
                      - ((!P.closure && !P.backward && !P.counting) ? (long long)tb_bytes(P, geo, MED_MODE_TB) : 0);
  int regFree = P.regBudget;                               // VGPRs for loop-invariant records (3 per record, +1 per round)
  if (P.Spad * 8 >= (1 << 16)) regFree = 0;
  for (MedRoundInfo &ri : P.roundInfo) for (MedSlotInfo &sl : ri.slots) { sl.place = MED_PLACE_GLOBAL; sl.ldsOff = 0; }
  auto toLds = [&](MedSlotInfo &sl) {
    const long long n = ntokT[sl.T] * LPG;
    if (n * 16 > ldsFree) return false;
    sl.place = MED_PLACE_LDS; sl.ldsOff = (long long)P.ldsImageIdx.size();
    for (long long k = 0; k < n; ++k) P.ldsImageIdx.push_back(sl.recBase + k);
    ldsFree -= n * 16;
    return true;
  };
  auto toReg = [&](MedRoundInfo &ri, size_t k) {
    // count programs: + one fp32 usage accumulator per record that takes part in the counting (levelled form: all; flat form: the usage pass)
    const int cost = ri.flat ? 4 : 3 + (k == 0 ? 1 : 0) + ((P.counting && !P.flatCount) ? 1 : 0);
    if (cost > regFree) return false;
    ri.slots[k].place = MED_PLACE_REG; regFree -= cost;
    return true;
  };
  if (env_int("MB_JIT_PLACE", 1) == 0) {   // legacy placement: token-independent records in LDS, everything else global
    for (MedRoundInfo &ri : P.roundInfo) for (MedSlotInfo &sl : ri.slots) if (sl.T == 3) toLds(sl);
    return;
  }
  // pass 1: input-token records -> VGPRs (they would otherwise be per-step global loads); output-token records -> LDS
  for (MedRoundInfo &ri : P.roundInfo)
    for (size_t k = 0; k < ri.slots.size(); ++k) {
      if (ri.slots[k].T == 1) toReg(ri, k);
      else if (ri.slots[k].T == 2) toLds(ri.slots[k]);
    }
  // pass 2: token-independent records -> VGPRs while the budget lasts, then LDS
  for (MedRoundInfo &ri : P.roundInfo)
    for (size_t k = 0; k < ri.slots.size(); ++k)
      if (ri.slots[k].T == 3 && !toReg(ri, k)) toLds(ri.slots[k]);
  P.regUsed = P.regBudget - regFree;
  // pass 3: leftovers (input-token records beyond the VGPR budget, match tables) -> LDS if they still fit
  for (MedRoundInfo &ri : P.roundInfo)
    for (MedSlotInfo &sl : ri.slots)
      if (sl.place == MED_PLACE_GLOBAL) toLds(sl);
}

// Spilled VGPRs of the (single) kernel in a code object: the value behind ".vgpr_spill_count" in the msgpack metadata
// note (a positive fixint, or 0xcc/0xcd/0xce + big-endian uint8/16/32).  -1 if the key is missing.
long long medium_jit_spill_count(const std::string &code) {
  static const char key[] = ".vgpr_spill_count";
  const size_t p = code.find(key);
  if (p == std::string::npos || p + sizeof(key) - 1 >= code.size()) return -1;
  const unsigned char *q = (const unsigned char *)code.data() + p + sizeof(key) - 1;
  const size_t left = code.size() - (p + sizeof(key) - 1);
  if (q[0] <= 0x7f) return q[0];
  if (q[0] == 0xcc && left >= 2) return q[1];
  if (q[0] == 0xcd && left >= 3) return ((long long)q[1] << 8) | q[2];
  if (q[0] == 0xce && left >= 5) return ((long long)q[1] << 24) | ((long long)q[2] << 16) | ((long long)q[3] << 8) | q[4];
  return -1;
}

std::string medium_jit_source(const mb_machine *m, const MedProgram &P, const MedGeom &geo, int mode, int matKind) {
  std::ostringstream defs, pre, body, post, flat;
  const int S = m->S;
  const bool counting = mode == MED_MODE_COUNT && !P.flatCount, tbmode = mode == MED_MODE_TB, maxmode = mode == MB_VITERBI || tbmode;
  const bool materialise = matKind == MED_MAT_FULL;
  const int threads = geo.waves * 64;
  defs << "#define JS " << S << "\n#define JSPAD " << P.Spad << "\n#define JNS " << P.NS << "\n#define JG " << P.G
       << "\n#define JC " << geo.C << "\n#define JWAVES " << geo.waves << "\n#define JMODE " << (maxmode ? 1 : (mode == MED_MODE_COUNT ? 2 : 0))
       << "\n#define JSTORE2 " << env_int("MB_JIT_STORE2", (S % 2 == 0 && P.LPG <= 8) ? 1 : 0)
       << "\n#define JSTORENT " << env_int("MB_JIT_STORENT", 0)
       << "\n#define JENV " << (geo.env ? 1 : 0) << "\n#define JMAT " << matKind << "\n#define JTB " << (tbmode ? 1 : 0) << "\n#define JTBS " << tb_lds_stride(P) << "\n#define JSB " << medium_tb_stride(S)
       << "\n#define JNH " << P.haloStates.size() << "\n#define JNHP " << std::max<size_t>(P.haloStates.size(), 1)
       << "\n#define JNHR " << std::max<size_t>((P.haloStates.size() + threads - 1) / threads, 1) << "\n#define JHALOT " << (materialise ? geo.haloSteps : 0)
       // NEIGHBOUR SYNCHRONISATION instead of a workgroup barrier per step (tiles without a matrix whose halo rows one wavefront moves):
       // see the skeleton.  Measured on psw2dna 64 x 487 x 2 kb: log-likelihood tiles 1 124 -> 1 188 G cells/s, traceback-byte Viterbi
       // 593 -> 617, the count sweep 224 -> 220 (its step is paced by the Backward loads, not by the barrier): off for that one
       // (round 5: on for the count sweep too when a wavefront is ONE column -- the 482-state machine: 87 -> 90 G lattice-cells/s)
       << "\n#define JNBSYNC " << ((env_int("MB_JIT_NEIGHBOUR_SYNC", (mode == MED_MODE_COUNT && P.G != 1) ? 0 : 1) && matKind == MED_MAT_ROLL && P.haloStates.size() <= 64 && geo.waves > 1 && geo.waves <= 16) ? 1 : 0)
       << "\n#define JFLAGOFF " << lds_payload_bytes(P, geo, mode)
       << "\n#define JBDIST " << (env_int("MB_JIT_B_DISTANCE", 1) == 2 ? 2 : 1)      // count sweep: steps the Backward supercells are fetched ahead (2: measured 218 vs 221 G lattice-cells/s -- the loads cost issue and LDS writes, not exposed latency)
       << "\n#define JDBG " << jit_debug_bits()      // experiments only (wrong results): 1 = no Backward loads, 2 = no halo loads
       << "\n#define JFLAT " << (P.flatCount ? 1 : 0) << "\n#define JNACC " << P.accEntries << "\n#define JNTRANS " << m->nTrans
       << "\n#define JNALL " << P.accAllEntries << "\n#define JNLOOP " << std::max(0, P.accEntries - P.LPG) << "\n#define JACCOFF " << (P.counting ? acc_offset(P, geo) : 0)
       << "\n#define JNOUT " << m->nOut << "\n#define JENDNODE " << P.dev.endNode
       << "\n#define JCR " << (geo.compact ? 1 : 0) << "\n#define JKC " << medium_compact_len(P) << "\n#define JRINGD " << ring_doubles(P, geo)
       << "\n#define JLDSRECS " << (long long)P.ldsImageIdx.size() << "\n#define JTOKW " << P.tokWindow
       << "\n#define JTOKN " << (P.tokWindow + geo.C - 1 + threads - 1) / threads
       << "\n#define JDUMMYOFF " << P.dummyOff << "\n#define JHALO " << (S + threads - 1) / threads << "\n";
  defs << "__device__ const int jHaloState[] = {";   // states whose values cross a strip boundary (halo rows of JMAT == 2)
  for (size_t k = 0; k < std::max<size_t>(P.haloStates.size(), 1); ++k) defs << (k ? "," : "") << (k < P.haloStates.size() ? P.haloStates[k] : 0);
  defs << "};\n#define JHSTATE(k) jHaloState[k]\n";
  static const char *vec[4] = {"aDiag", "aLeft", "aDown", "aCur"};
  static const char *tok[4] = {"tokM16", "itOff16", "otOff16", "q16"};
  // STAGES.  The rounds between two synchronisation points do not depend on one another (a round's same-cell candidates read
  // what EARLIER stages wrote; build_program), but each ends with an LDS store to an address the compiler cannot tell from the
  // next round's LDS loads, so straight-line "load, fold, store" per round runs the rounds of a stage one after the other -- a
  // full LDS round trip + the log-sum-exp chain each (psw2dna's count sweep: 11 rounds in 3 stages, ~400 cycles per round of a
  // 6 900-cycle step whose vector instructions need 1 500).  A stage is therefore emitted as: every load of every round (records
  // placed in LDS, source values), then the folds, then the stores; rounds too long for that (> JIT_MAX_CANDS candidates) keep
  // their running form and go between the loads and the folds of the others.  MB_JIT_STAGE_LOADS=0: round by round.
  // A stage's loads are issued in batches of at most MB_JIT_STAGE_MAXLOADS source values (two VGPRs each until folded).
  const bool stageLoads = env_int("MB_JIT_STAGE_LOADS", 1) != 0;
  const int stageMaxLoads = std::max(1, env_int("MB_JIT_STAGE_MAXLOADS", 32));
  int stagePending = 0;
  bool firstStageDone = false;
  std::ostringstream sLoad, sBig, sFold;
  auto flushStage = [&]() {
    stagePending = 0;
    if (sLoad.str().empty() && sBig.str().empty() && sFold.str().empty()) return;
    body << "      {\n" << sLoad.str() << sBig.str() << sFold.str() << "      }\n";
    sLoad.str(""); sBig.str(""); sFold.str("");
  };
  for (size_t r = 0; r < P.roundInfo.size(); ++r) {
    const MedRoundInfo &ri = P.roundInfo[r];
    const int n = (int)ri.slots.size();
    const std::string R = "_" + std::to_string(r);
    const bool big = !tbmode && !ri.flat && n > JIT_MAX_CANDS;
    // name of the record of slot k; emits its load (loop-invariant ones go to the prologue)
    std::ostringstream &out = ri.flat ? flat : (big ? sBig : sLoad);
    auto rec = [&](int k) {
      const MedSlotInfo &sl = ri.slots[k];
      const std::string name = "r" + std::to_string(r) + "_" + std::to_string(k);
      if (sl.place == MED_PLACE_REG)
        pre << "  const Rec " << name << " = ld_g(grb + " << sl.recBase * 16 << "ull, " << (sl.T == 1 ? "itOff16" : "q16") << ");\n";
      else if (sl.place == MED_PLACE_LDS)
        out << "        const Rec " << name << " = ld_l(ldsRec, " << sl.ldsOff * 16 << "u + " << tok[sl.T] << ");\n";
      else if ((long long)P.rec.size() * 16 < (1ll << 31))   // buffer load: lane offset in a VGPR, slot base in an SGPR -> no loop-invariant 64-bit address per slot
        out << "        const Rec " << name << " = ld_b(recRsrc, " << tok[sl.T] << ", " << sl.recBase * 16 << ");\n";
      else
        out << "        const Rec " << name << " = ld_g(grb + " << sl.recBase * 16 << "ull, " << tok[sl.T] << ");\n";
      return name;
    };
    if (ri.flat) {
      // usage pass of a flat count program (mb_medium.hip, append_flat_usage): every slot is one transition per lane,
      // term = exp((F(src) + w) + (B(dst) - LL)); bvec holds B - LL.  Loop-invariant records sum in a register (fp32, one tile of
      // steps) and reach the workgroup's LDS accumulator after the step loop -- their accumulator offset is read again there
      // instead of living in a VGPR --, the others add to LDS per step.
      flushStage();
      // The slots go in batches of MB_JIT_FLAT_CHUNK (24): the loads of a batch are in flight together, and a program of a hundred
      // slots (16 columns per wavefront on a dense machine) no longer keeps every record and term live at once (600+ spilled VGPRs).
      std::vector<std::string> fn(n);
      const int chunk = env_int("MB_JIT_FLAT_CHUNK", 24) > 0 ? env_int("MB_JIT_FLAT_CHUNK", 24) : n;
      flat << "      if (JINSIDE && !(JDBG & 8)) {  // usage pass: " << n << " slot(s); lanes of columns outside the lattice (stale ring values) sit it out\n";
      for (int k0 = 0; k0 < n; k0 += chunk) {
        const int k1 = std::min(n, k0 + chunk);
        if (n > chunk) flat << "        {\n";
        for (int k = k0; k < k1; ++k) fn[k] = rec(k);
        for (int k = k0; k < k1; ++k)
          flat << "        const double x" << k << " = (med_lds(ldsb, " << vec[ri.slots[k].T] << " + (int)(" << fn[k] << ".srcOff & 0xFFFFu)) + " << fn[k] << ".w) + med_lds(ldsb, aB + (int)(" << fn[k] << ".srcOff >> 16));\n";
        for (int k = k0; k < k1; ++k) {
          const MedSlotInfo &sl = ri.slots[k];
          const std::string term = "ex2(x" + std::to_string(k) + ")";
          if (sl.place == MED_PLACE_REG) {
            pre << "  float acc_" << fn[k] << " = 0.0f;\n";
            flat << "        acc_" << fn[k] << " += " << term << ";\n";
            post << "  cnt_flush(ldsb, accBase2 + ld_g(grb + " << sl.recBase * 16 << "ull, " << (sl.T == 1 ? "itOff16" : "q16") << ").dstOff, acc_" << fn[k] << ");\n";
          } else
            flat << "        cnt_flush(ldsb, accBase + " << fn[k] << ".dstOff, " << term << ");\n";
        }
        if (n > chunk) flat << "        }\n";
      }
      flat << "      }\n";
      continue;
    }
    // count mode (levelled form): usage term exp(v_k + bl) of candidate k, bl = B(dst) - logLike (src/backward.cpp:58-87).  The
    // log-sum-exp of the state already formed e_k = exp(v_k - gM) for every candidate, so the usage is e_k * s with ONE more
    // exponential per state (per group of JIT_MAX_CANDS on the running-max path), s = exp(max + bl) -- `term` is that product, or
    // the lone candidate's own exp(v_0 + bl).  A record that sits in VGPRs names the same transition for the whole sweep, so its
    // usage is summed in a register (fp32, at most one tile of steps) and reaches the workgroup's LDS accumulator once, after the
    // step loop; the others add to LDS per step (ds_add_f64).
    auto countTerm = [&](std::ostringstream &o, const MedSlotInfo &sl, const std::string &name, const std::string &term) {
      if (sl.place == MED_PLACE_REG) {
        pre << "  float acc_" << name << " = 0.0f;\n";
        o << "        acc_" << name << " += " << term << ";\n";
        post << "  cnt_flush(ldsb, accBase2 + (" << name << ".srcOff >> 16), acc_" << name << ");\n";
      } else
        o << "        cnt_flush(ldsb, accBase + (" << name << ".srcOff >> 16), " << term << ");\n";
    };
    auto V = [&](int k) { return "v" + R + "_" + std::to_string(k); };
    auto E = [&](int k) { return "e" + R + "_" + std::to_string(k); };
    std::vector<std::string> nm(n);
    // (in-place ring: the first stage's emit rounds read the column's vector as "the step before" and overwrite it -- every load of the stage
    //  goes in front of its first store, whatever the batch size; MedProgram::inPlaceOk bounds the stage's slots)
    if (!big) { if (stagePending > 0 && stagePending + n > stageMaxLoads && !(geo.compact && !firstStageDone)) flushStage(); stagePending += n; }
    sFold << "        // round " << r << ": " << n << " candidate slot(s)\n";
    if (tbmode) {
      // max semiring with the index of the FIRST maximal candidate: slots come table by table (match, input-only, output-only,
      // same-cell) and inside a table in the reference's list order, which is its enumeration order (src/dpmatrix.defs.h:93-103);
      // strict > keeps the first maximum like std::max_element.  code = table << 6 | index in the table's list.
      int perT[4] = {0, 0, 0, 0};
      for (int k = 0; k < n; ++k) nm[k] = rec(k);
      for (int k = 0; k < n; ++k)
        sLoad << "        const double " << V(k) << " = med_lds(ldsb, " << vec[ri.slots[k].T] << " + SRCOFF(" << nm[k] << ".srcOff)) + " << nm[k] << ".w;\n";
      sFold << "        double res" << R << " = " << V(0) << "; unsigned x" << R << " = " << ((ri.slots[0].T << 6) | 0) << "u;\n";
      perT[ri.slots[0].T] = 1;
      for (int k = 1; k < n; ++k) {
        const int code = (ri.slots[k].T << 6) | perT[ri.slots[k].T]++;
        sFold << "        { const bool g = " << V(k) << " > res" << R << "; res" << R << " = g ? " << V(k) << " : res" << R << "; x" << R << " = g ? " << code << "u : x" << R << "; }\n";
      }
      sFold << "        const unsigned dW" << R << " = active ? " << nm[0] << ".dstOff : (unsigned)JDUMMYOFF;\n";
      sFold << "        const int dOff" << R << " = (int)DSTOFF(dW" << R << ");\n";
      sFold << "        *(double *)(ldsb + (aCur + dOff" << R << ")) = JCLIP(res" << R << ");\n";
      sFold << "        JCSTORE(dW" << R << ", JCLIP(res" << R << "));\n";
      sFold << "        tbCol[dOff" << R << " >> 3] = (unsigned char)x" << R << ";\n";
    } else if (!big) {
      for (int k = 0; k < n; ++k) nm[k] = rec(k);
      for (int k = 0; k < n; ++k)
        sLoad << "        const double " << V(k) << " = med_lds(ldsb, " << vec[ri.slots[k].T] << " + SRCOFF(" << nm[k] << ".srcOff)) + " << nm[k] << ".w;\n";
      if (counting)   // posterior usage of every candidate's transition: exp(F(src) + w + B(dst) - LL), src/backward.cpp:58-87
        sLoad << "        const double bl" << R << " = med_lds(ldsb, aB + (int)(JINSIDE ? DSTOFF(" << nm[0] << ".dstOff) : (unsigned)JDUMMYOFF)) + negLL;   // the dummy entry of bvec holds -inf\n";
      if (ri.fused)   // flat count program: the round's emitting candidates are real transitions into the lane's state -- B(state) - LL once per round (bvec holds B - LL)
        sLoad << "        const double bf" << R << " = med_lds(ldsb, aB + (int)(JINSIDE ? (" << nm[0] << ".dstOff >> 16) : (unsigned)JDUMMYOFF));\n";
      if (n == 1) {
        sFold << "        const double res" << R << " = " << V(0) << ";\n";
      } else {
        sFold << "        double mx" << R << " = dmax(" << V(0) << ", " << V(1) << ");\n";
        for (int k = 2; k < n; ++k) sFold << "        mx" << R << " = dmax(mx" << R << ", " << V(k) << ");\n";
        if (maxmode) sFold << "        const double res" << R << " = mx" << R << ";\n";
        else {
          sFold << "        const double gM" << R << " = (mx" << R << " == NEG_INF) ? 0.0 : mx" << R << ";\n";
          for (int k = 0; k < n; ++k) sFold << "        const float " << E(k) << " = ex2(" << V(k) << " - gM" << R << ");\n";
          sFold << "        const float sm" << R << " = " << E(0);
          for (int k = 1; k < n; ++k) sFold << " + " << E(k);
          sFold << ";\n        const double res" << R << " = gM" << R << " + (double)(__builtin_amdgcn_logf(sm" << R << ") * MED_LN2);\n";
        }
      }
      sFold << "        { const unsigned dW = active ? " << nm[0] << ".dstOff : (unsigned)JDUMMYOFF; *(double *)(ldsb + (aCur + (int)DSTOFF(dW))) = JCLIP(res" << R << "); JCSTORE(dW, JCLIP(res" << R << ")); }\n";
      if (ri.fused && !(jit_debug_bits() & 16))
        for (int k = 0; k < n; ++k) {
          const MedSlotInfo &sl = ri.slots[k];
          if (sl.T >= 3) continue;
          const std::string term = "ex2(" + V(k) + " + bf" + R + ")";
          if (sl.place == MED_PLACE_REG) {      // the same transition for the whole sweep of the column: summed in a register, set down after the step loop
            pre << "  float acc_" << nm[k] << " = 0.0f;\n";
            sFold << "        acc_" << nm[k] << " += " << term << ";\n";
            post << "  cnt_flush(ldsb, accBase2 + (" << nm[k] << ".srcOff >> 16), acc_" << nm[k] << ");\n";
          } else
            sFold << "        cnt_flush(ldsb, accBase + (" << nm[k] << ".srcOff >> 16), " << term << ");\n";
        }
      if (counting) {
        if (n == 1) countTerm(sFold, ri.slots[0], nm[0], "ex2(" + V(0) + " + bl" + R + ")");
        else {
          sFold << "        const float sB" << R << " = ex2(mx" << R << " + bl" << R << ");   // (mx, not gM: no candidate at all -> exp(-inf) = 0, whatever B - logLike is)\n";
          for (int k = 0; k < n; ++k) countTerm(sFold, ri.slots[k], nm[k], E(k) + " * sB" + R);
        }
      }
    } else {
      // many candidates: groups of JIT_MAX_CANDS folded into a running (max, scaled sum)
      sBig << "        {  // round " << r << ": " << n << " candidate slot(s)\n";
      sBig << "        double accM = NEG_INF; float accS = 0.0f; unsigned dstOff = 0xFFFFFFFFu;\n";
      for (int k0 = 0; k0 < n; k0 += JIT_MAX_CANDS) {
        const int k1 = std::min(n, k0 + JIT_MAX_CANDS);
        sBig << "        {\n";
        for (int k = k0; k < k1; ++k) nm[k] = rec(k);
        if (k0 == 0) sBig << "        dstOff = " << nm[0] << ".dstOff;\n";
        for (int k = k0; k < k1; ++k)
          sBig << "        const double v" << k << " = med_lds(ldsb, " << vec[ri.slots[k].T] << " + SRCOFF(" << nm[k] << ".srcOff)) + " << nm[k] << ".w;\n";
        sBig << "        double mx = v" << k0 << ";\n";
        for (int k = k0 + 1; k < k1; ++k) sBig << "        mx = dmax(mx, v" << k << ");\n";
        if (maxmode) sBig << "        accM = dmax(accM, mx);\n";
        else {
          sBig << "        const double nm = dmax(accM, mx), gM = (nm == NEG_INF) ? 0.0 : nm;\n";
          for (int k = k0; k < k1; ++k) sBig << "        const float e" << k << " = ex2(v" << k << " - gM);\n";
          sBig << "        accS = accS * ex2(accM - gM)";
          for (int k = k0; k < k1; ++k) sBig << " + e" << k;
          sBig << ";\n        accM = nm;\n";
          if (counting) {
            sBig << "        const float sB = ex2(nm + (med_lds(ldsb, aB + (int)(JINSIDE ? DSTOFF(dstOff) : (unsigned)JDUMMYOFF)) + negLL));\n";
            for (int k = k0; k < k1; ++k) countTerm(sBig, ri.slots[k], nm[k], "e" + std::to_string(k) + " * sB");
          }
        }
        sBig << "        }\n";
      }
      if (maxmode) sBig << "        const double res = accM;\n";
      else sBig << "        const double res = ((accM == NEG_INF) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);\n";
      sBig << "        { const unsigned dW = active ? dstOff : (unsigned)JDUMMYOFF; *(double *)(ldsb + (aCur + (int)DSTOFF(dW))) = JCLIP(res); JCSTORE(dW, JCLIP(res)); }\n";
      sBig << "        }\n";
    }
    if (ri.sync || (!stageLoads && !(geo.compact && !firstStageDone))) flushStage();
    if (ri.sync) { body << "      med_wave_sync();\n"; firstStageDone = true; }
  }
  flushStage();
  std::string src = kMedJitSkeleton;
  auto replace = [&](const std::string &mark, const std::string &with) {
    const size_t p = src.find(mark);
    if (p != std::string::npos) src.replace(p, mark.size(), with);
  };
  replace("/*@DEFS@*/", defs.str());
  replace("/*@PRE@*/", pre.str());
  replace("/*@BODY@*/", body.str());
  replace("/*@FLAT@*/", flat.str());
  replace("/*@POST@*/", post.str());
  return src;
}

bool medium_jit_get(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, int mode, int matKind, bool allowReplan) {
  MedGeom geo = geoIn;
  if (matKind != MED_MAT_FULL) geo.haloSteps = 0;   // only the matrix kernel pre-loads a tile's halo supercells
  const bool materialise = matKind == MED_MAT_FULL;
  MedJit &J = P.jit[medium_jit_slot(mode, matKind, geo.level, geo.env)];
  if (J.tried) return J.func != nullptr;
  if ((mode == MED_MODE_COUNT) != P.counting) return false;   // count programs carry packed records: one mode only
  J.tried = true;
  const char *e = opt_env("MB_MEDIUM_JIT");
  if (e && *e == '0') return false;
  long long totalSlots = 0;
  for (const MedRoundInfo &ri : P.roundInfo) {
    if (ri.slots.empty()) return false;
    totalSlots += (long long)ri.slots.size();
  }
  if (totalSlots > 20000) return false;
  if (P.roundInfo.size() > 4096) return false;   // keep the generated code within reach of the instruction cache
  std::string code, src;
  bool fromCache = false;
  for (int attempt = 0; attempt < 8; ++attempt) {
    J.ldsBytes = medium_jit_lds_bytes(P, geo, mode);
    if (J.ldsBytes > 160 * 1024) {
      if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] jit (mode %d, matrix kind %d): %zu bytes of LDS at register budget %d -- ahead-of-time kernel\n", mode, matKind, (size_t)J.ldsBytes, P.regBudget);
      return false;
    }
    src = medium_jit_source(m, P, geo, mode, matKind);
    if (const char *dump = opt_env("MB_MEDIUM_JIT_DUMP")) {
      if (FILE *f = fopen((std::string(dump) + (mode == MB_VITERBI ? ".vit" : (mode == MED_MODE_COUNT ? ".cnt" : (mode == MED_MODE_TB ? ".tb" : ".sum"))) + (materialise ? ".mat" : (matKind == MED_MAT_ROLL ? ".tiles" : ".roll")) + (P.backward ? ".bwd" : ".fwd") + (P.closure ? ".clos" : ".exact") + ".hip").c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    std::string log;
    if (!jit_compile(src, "mb_medium_jit.hip", code, &log, &fromCache)) {
      if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] hiprtc failed:\n%s\n", log.c_str());
      return false;
    }
    // What counts is SCRATCH memory (.private_segment_fixed_size): registers parked in AGPRs (a workgroup of <= 4 wavefronts has 512
    // registers per lane) are reported as spilled VGPRs too -- the ROCm 7.0 compiler PyTorch bundles does that for the 482-state count
    // kernel at every budget -- and cost nothing.
    long long spills = medium_jit_spill_count(code);
    if (jit_kernel_meta(code, ".private_segment_fixed_size") == 0) spills = 0;
    if (attempt == 0 && (jit_debug_bits() & 4)) spills = 3;      // experiments: force one re-plan
    if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] jit %s: register budget %d, %lld spilled VGPRs (scratch %lld bytes)\n", mode == MB_VITERBI ? "max" : (mode == MED_MODE_COUNT ? "count" : (mode == MED_MODE_TB ? "max+tb" : "sum")), P.regBudget, spills, jit_kernel_meta(code, ".private_segment_fixed_size"));
    // (the LAST attempt keeps what it compiled: re-planning behind it would leave the placement -- the LDS image, the record tables
    // the host refreshes -- one step ahead of the code; round 4 found exactly that, counts of 1e19, when a kernel never stopped spilling)
    if (spills > 0 && !allowReplan) return false;      // (a trial build that must not touch the program's placement: medium_roll_geometry)
    if (spills <= 0 || P.regBudget == 0 || attempt == 7) break;
    // the compiler ran out of VGPRs: move records from registers to LDS / global and regenerate.  The placement is
    // shared by both semirings of this program, so a kernel already built for the other one is dropped.
    P.regBudget = std::max(0, std::min(P.regBudget, P.regUsed) - std::max(9, (int)spills + 3));   // cut from what the plan really spent
    // (flat count programs: which usage records are loop-invariant changes with the placement, and with it the size of the loop-time
    //  accumulator table the placement has to leave room for -- planned once against the largest table, then against the real one)
    if (P.counting && P.flatCount) { P.accEntries = (int)m->nTrans + P.LPG; medium_jit_plan(m, P, geoIn); medium_count_layout(m, P); }
    medium_jit_plan(m, P, geoIn);
    medium_count_layout(m, P);
    if (!medium_refresh_weights(m, P)) return false;
    for (MedJit &O : P.jit) {
      if (&O == &J) continue;
      if (O.module) (void)hipModuleUnload((hipModule_t)O.module);
      O = MedJit();
    }
  }
  // A kernel that keeps scratch memory at the last budget is compiled once more with the VGPR -> AGPR spill path of the compiler
  // switched off.  Round 4, ROCm 7.2: the 126-slot usage pass of a flat count program as ONE batch (618 spilled VGPRs, 512 registers
  // in use) lost the Backward values of one group of four states -- same source right at -O1 and with this option, wrong at -O3;
  // in the ISA the value travels a248 -> a219 through v_accvgpr_mov between its spill and its reload.  The pass is batched now
  // (MB_JIT_FLAT_CHUNK) and no kernel of the test suite or the bench ends here; MB_JIT_AGPR_SPILLS=1 keeps the compiler's default.
  struct MoreOpts { bool on; explicit MoreOpts(bool o) : on(o) { if (on) jit_more_opts("-mllvm -amdgpu-spill-vgpr-to-agpr=0"); } ~MoreOpts() { if (on) jit_more_opts(nullptr); } };
  const MoreOpts noAgprSpills(jit_kernel_meta(code, ".private_segment_fixed_size") > 0 && env_int("MB_JIT_AGPR_SPILLS", 0) == 0);
  if (noAgprSpills.on) {
    std::string log;
    if (!jit_compile(src, "mb_medium_jit.hip", code, &log, &fromCache)) return false;
    if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] jit: scratch memory at the last budget -- compiled again without AGPR spill slots (scratch %lld bytes)\n", jit_kernel_meta(code, ".private_segment_fixed_size"));
  }
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
    // a cached code object the loader rejects (truncated file, other compiler build): drop it and compile afresh, once
    std::string log;
    mod = nullptr;
    if (fromCache) { jit_evict(src); if (!jit_compile(src, "mb_medium_jit.hip", code, &log, nullptr) || hipModuleLoadData(&mod, code.data()) != hipSuccess) mod = nullptr; }
    if (!mod) return false;
  }
  if (hipModuleGetFunction(&fn, mod, "k_medium_jit") != hipSuccess) { (void)hipModuleUnload(mod); return false; }
  (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  J.module = mod; J.func = fn;
  return true;
}

void medium_jit_free(MedProgram &P) {
  for (MedJit &J : P.jit) {
    if (J.module) (void)hipModuleUnload((hipModule_t)J.module);
    J = MedJit();
  }
}

}  // namespace mb

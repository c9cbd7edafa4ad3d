// mb_wide.h -- "one workgroup = one column" kernel family for large one-tape machines (see mb_wide.hip).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "mb_internal.h"

namespace mb {

// One candidate of one lane (16 bytes, one global_load_dwordx4): value = vector[src] + w.
//   src = sel << 30 | index   sel 0: this column's state vector, 1: this column's extra entries (emit-only parts of
//                             states that also have silent predecessors, the dummy entry), 2: the previous column's
//                             state vector (index S = -inf sentinel, S + 1 = seed).
struct alignas(16) WideRec {
  double w;
  uint32_t src;
  uint32_t pad;
};

// A round: every lane group of one workgroup finalises one node (state or emit-only part).  Lane `l` of a group of g
// lanes folds candidates l, l + g, l + 2g, ... of its node (depth slots), the group is reduced with shuffles, its first
// lane stores.  (Host-side planning structure; the kernel reads the linearised streams below.)
struct WideRound {
  int recBase, tokStride, depth, dstBase;
  int maxG, sync, pad0, pad1;
};

// What the kernel reads: the slots of one column as a flat stream of [slot][lane] records, so that a lane's loads are
// known ahead and kept WIDE_RING deep in flight across rounds, barriers and columns.  Segment A (the rounds up to the
// last one with emitting candidates) exists once per output token, segment B (silent closure) is shared by all tokens.
// WideRec::pad of slot j: bit 31 = the round ends with this slot, bit 30 = __syncthreads() after it (both uniform over
// the lanes); in the last slot of a round, per lane: bit 29 = destination is an extra entry, bits 28..26 = log2 of the
// lane-group size, bits 25..0 = destination index (all ones: this lane stores nothing).
constexpr int WIDE_RING = 8;
struct WideDev {
  const WideRec *segA, *segB;
  long long strideA;   // records per token table of segment A (= nA * W)
  int nA, nB;          // slots; nA + nB is a multiple of WIDE_RING
  int S, NV, NX, W;
  int resultIdx;       // state whose value in the last column is the log-likelihood
  int backward;
  int inputTape;       // 1: the machine is a recogniser, the column index is the input position
  int lastOnly;        // 1: `pool` holds ONE column per pair (cellBase = its offset): only the last column of the sweep is stored
};

// Forward / Backward (log-sum-exp) run in single precision RELATIVE TO A PER-COLUMN fp64 REFERENCE: a column's vector
// holds x - R with R the running sum of the column maxima, so the entries that carry the probability mass are near 0
// where fp32 resolves 1e-7, and a cell of the materialised matrix is R + (double)entry.  Stream entry = 8 bytes
// (weight as float, 16-bit source indices for both column parities); slot flags (uniform) come from a byte table read
// with scalar loads; the per-lane destination word of a round is the `src` of a control entry that follows the round.
struct WideRec32 { float w; uint32_t src; };
enum { WIDE_F_END = 1, WIDE_F_SYNC = 2, WIDE_F_CTRL = 4, WIDE_F_PREV = 8 };   // PREV: the slot reads the previous column (hybrid layout)
struct WideDev32 {
  const WideRec32 *segA, *segB;
  const unsigned long long *flags;   // 8 slot-flag bytes per word
  long long strideA;
  int nA, nB;
  int S, NV, NX, W;
  int resultIdx, backward;
  int inputTape;
  int lastOnly;
};

// Levelled max programs (Viterbi: one rounded add per transition, the silent levels in the reference's order) of profile-like
// machines are hundreds of THIN levels -- a dozen states each, all in the first wavefront -- between a few wide rounds.  Walking
// them with all 16 wavefronts through W-lane record slots made the sweep stream 6 MB of padding records per column and
// workgroup (bandwidth-bound at 3.6 G cells/s where the log-sum-exp sweep, which closes the levels on the host, ran 16).
// k_wide_viterbi walks a column as a list of PHASES: a wide phase is a run of rounds every wavefront works on (W-lane slots);
// a thin phase is a run of rounds whose nodes all sit in the first wavefront -- 64-lane slots, read by that wavefront alone
// with a deep prefetch ring while the others wait at the barrier that ends the phase.  Same rounds, same candidates, same
// order as the generic kernel: the results are the same bits.
constexpr int WIDE_THIN_RING = 16;
struct WidePhase { int thin, inA, nSlots, off; };      // off: first slot of the phase within its stream (wide / thin, A / B)
struct WideVitDev {
  const WideRec *wideA, *wideB, *thinA, *thinB;        // A: one table per output token (the emitting rounds), B: shared
  long long strideWideA, strideThinA;                  // records per token table
  const WidePhase *phase;
  int nPhases;
};

// ---- RETIMED sweep (k_wide_retimed) --------------------------------------------------------------------------------------------
// A column of a levelled one-tape program is a chain of hundreds of dependent silent levels, and columns follow one another -- but
// the dependency graph over (column, state) is looser than that: a cell only waits for its own predecessors.  Give every
// state s a time offset tau(s) and let cell (c, s) be computed at time step c * P + tau(s).  The schedule is valid iff
//     tau(s) >= tau(u) + 1      for every silent transition    u -> s   (same column), and
//     tau(s) + P >= tau(u) + 1  for every emitting transition  u -> s   (u in the column before);
// the smallest feasible period P is the machine's largest (transitions per emission) over its cycles -- 9 for the 20-node
// profile . simple_introns . translate . dnapsw machine whose columns are 366 levels deep.  Step t finalises every state with
// tau = t (mod P), each for ITS OWN column (t - tau) / P: P wide rounds per period instead of hundreds of thin levels, with
// tauMax / P columns in flight.  Every candidate is still ONE rounded add of one transition's weight and every state one maximum
// (or one log-sum-exp) over its direct predecessors: the cells are the levelled program's, bit for bit in the max semiring.
// The kernel sees a PERIOD as a flat list of rounds (one stage per residue of tau mod P, a barrier after each); a node carries
// ktau = tau / P, the number of columns it lags the period's newest column.  Values live in a ring of NB column vectors of NVs
// doubles: the S states, two constants (-inf, 0.0 = the seed), and RELAY entries -- a source some reader sees NB * P or more
// steps after it was written is copied (x + 0.0) every NB * P - 1 steps into an entry of its own, and the late reader takes the
// youngest copy.  Emitting transitions are candidates of their own; whether the token they emit is the one of the lane's column
// (and whether the column is the first one, for the seed) comes from a table of penalties (0 / -inf) per (ktau, token), rebuilt
// every period one period ahead by the first lanes of the workgroup from a 64-entry window of the sequence.
//   src = penalty entry byte offset << 16 | a0        a0: ring entry of the source when the newest column sits in vector 0,
//           ((- ktau - emits) mod NB) * NVs + index; the kernel adds the rotation of the period and wraps once
//   pad (last slot of a round) = END << 31 | SYNC << 30 | MIXED << 29 | log2(group) << 26 | ktau << 20 | ((- ktau) mod NB) << 18 | entry
//           MIXED (first lane of a wavefront): the wavefront holds lane groups of different sizes -- the reduction steps are masked
constexpr int WIDE_RET_TOKWIN = 64;
constexpr uint32_t WIDE_RET_NO_DST = 0x3ffffu;
struct WideRetDev {
  const WideRec *rec;          // [nSlots + WIDE_RING][W]: the rounds of one period, then its first WIDE_RING slots again
  int nSlots;                  // multiple of WIDE_RING (of a part's own ring depth, 4 or 8)
  int NB, NVs;                 // ring depth, doubles per ring vector (S + 2 + relays)
  int kMax;                    // largest ktau: a sequence of L columns takes L + 1 + kMax periods
  int rowLen, nPen;            // penalty table: rowLen = tokens + 2 entries (silent, each token, seed) per ktau, nPen entries in all
};

// ---- the sweep GENERATED per machine (mb_wide_jit.h / mb_wide_jit.cpp): a compiled module and the per-lane constant tables of its parts ----
constexpr int WIDE_JIT_MAX_PARTS = 16;
struct WideJitArgs {
  const uint32_t *tab[WIDE_JIT_MAX_PARTS];      // per part: [word][lane] constants (weights, source / destination addresses per ring rotation, lags, offsets)
  const uint32_t *impIdx[WIDE_JIT_MAX_PARTS];   // per part: exchange column of import i
  const uint32_t *stream[WIDE_JIT_MAX_PARTS];   // per part: [rotation][item][lane] packed address words of a streamed program (nullptr: all in registers)
  int nSeq, nExpTot;
  double *X;
  const long long *xOff;
  unsigned *err;
  long long timeoutTicks;
};
struct WideJitModule { void *module = nullptr, *func = nullptr; ~WideJitModule(); };
struct WideJitKernel {
  bool tried = false;
  std::shared_ptr<WideJitModule> mod;           // shared between programs of the same structure (a weight update finds its kernel again)
  size_t ldsBytes = 0;
  int W = 0, k = 0;
  std::vector<uint32_t *> d_tab, d_stream;
  WideJitArgs args{};
  std::string why;                              // why the interpreter kept the program (empty: built)
  void release();
};

// ---- k WORKGROUPS PER SEQUENCE (round 5; k_wide_retimed_parts) -----------------------------------------------------------------
// One workgroup per sequence leaves most of the chip idle when a batch has fewer sequences than the device has CUs (BASELINE config 5:
// 64 sequences, 256 CUs -- and 8 per GPU when the batch is split over eight).  The states of the machine are cut into k PARTS along a
// topological order of the strongly connected components of its transition graph (emitting transitions included, in the direction of
// the sweep): no cycle crosses a cut, so values flow from part p to parts > p only, and a consumer may lag its producer by any number
// of columns.  Every part is a retimed program of its own (own period, own ring, own record stream) over
//     its own states | the two constants | IMPORT nodes | EXPORT nodes | relays.
// An export node is a silent copy (x + 0.0) of a state some later part reads; its result goes to the EXCHANGE buffer
// X[sequence][column][export] (fp64, one 8-byte store of agent scope) instead of the matrix.  An import node's only candidate is
// `0.0 + (0.0 + X[column][export])`: the value enters through the PENALTY TABLE -- the table of a period is written one period ahead by
// the workgroup's first lanes; its last lanes append one entry per import -- so the record format and the slot loop are those of the
// one-workgroup sweep.  There are no flags: X is preset to all-ones (a NaN no cell ever holds) and a consumer lane re-reads its entry until
// it is something else (issued a period ahead, so the wait is normally over when it is looked at).  Workgroup ids are part-major: a
// consumer waits only for workgroups with lower ids, which the dispatcher starts first -- no deadlock even when the grid exceeds the chip --
// and every wait is bounded (WidePartArgs::timeoutTicks: the kernel raises *err, stops waiting and the host fails the call).
// Cells, traceback codes and log-likelihoods are those of the one-workgroup program, bit for bit in the max semiring (same candidates in
// the same order; the two copies add 0.0).
// A part WITH TWO-TRANSITION CANDIDATES (ret_merge in mb_wide.hip: silent transitions merged into their successors' candidate lists
// halve the stages of a period) has a second stream behind its records, w2Offset bytes into `ret.rec`: one double per lane and slot, the
// SECOND weight of the candidate (0.0 for a one-transition candidate) -- `(V(u) + (w + penalty)) + w2`.
struct WidePartDev {
  WideRetDev ret;              // nPen: the (ktau, token) entries only; the imports' entries follow them; rec: [slot][lane] of 32 bytes
  const uint32_t *gmap;        // [Sloc]: machine state of local state x (column of its matrix cell / traceback code)
  const uint32_t *impIdx;      // [nImp]: exchange column of import i
  int Sloc, nImp;              // own states (ring entries 0 .. Sloc - 1; Sloc, Sloc + 1: -inf and 0.0); imports
  int expBase, expIdx0, nExp;  // ring entries >= expBase are exports: entry expBase + j is exchange column expIdx0 + j
  int resultEntry;             // ring entry of the state whose last-column value is the log-likelihood, -1: another part has it
  int w2Offset;                // byte offset of the second weights' stream (parts with two-transition candidates)
};
struct WidePartArgs {
  const WidePartDev *parts;
  int nSeq, nExpTot;           // workgroup = part * nSeq + sequence; exchange columns per sequence column
  double *X;                   // [xOff[seq] + column][nExpTot]
  const long long *xOff;       // first exchange row of every sequence
  unsigned *err;               // raised by a lane whose wait ran out
  long long timeoutTicks;      // of wall_clock64() (100 MHz)
};
// what a build of a cut chose -- lanes, ring depth, the two-transition candidates of every part (indices into the part's edge list, with its
// length as a check) -- kept by the program across weight updates: the choice depends on the machine's graph, not on its weights
struct WidePartHint { bool valid = false, merge = true; int kWanted = 0, lanesAsked = 0, ringAsked = 0, W = 0, ring = 8; std::vector<std::vector<int>> merged; std::vector<size_t> nEdges; std::vector<int> period, periodMin; };
struct WidePartSet {
  bool ok = false;
  int kWanted = 0, lanesAsked = 0, ringAsked = 0;      // what was asked for (0: the builder's choice)
  int k = 0, W = 0, nExpTot = 0, ring = WIDE_RING;
  double modelCost = 0.0;                    // modelled ms per 10 000 columns of the slowest part (wide_parts_host's model)
  bool merge = false;                        // the parts carry two-transition candidates (kernel variant with the second weights' stream)
  size_t ldsBytes = 0;                       // largest part (with the fp64 correction term's table)
  std::vector<WideRec *> d_rec;
  std::vector<uint32_t *> d_tab;             // gmap + impIdx of every part, one allocation each
  WidePartDev *d_parts = nullptr;
  std::vector<WidePartDev> h_parts;          // (device pointers inside)
  std::vector<int> period, slots, nSync;     // per part, for the log
  // traceback-code programs: the decode tables of the parts' own candidate lists (two-transition candidates change the places), joined
  int *d_tbOff = nullptr; uint32_t *d_tbEntry = nullptr; long long tbEntries = 0;
  // the generated kernel of this cut ([1]: with the fp64 correction term), built on first use from the host copies of the parts' streams
  std::vector<std::vector<WideRec>> h_stream;
  std::vector<std::vector<uint32_t>> h_tab;
  WideJitKernel jit[2];
};

// a second sweep fused into the same launch (workgroups >= nFirst run it): Forward and Backward of one batch side by side
struct WideSecond { unsigned nFirst; const PairDesc *pairs; double *pool; void *scratch; };

struct WideProgram {
  bool ok = false, dirty = true;
  bool backward = false, viterbi = false;
  int stages = 0;            // closure stages the silent levels were grouped into (0 = levelled, exact)
  int W = 1024;
  long long nPairs = 0;      // closure pairs
  long long slotsPerColumn = 0, candsPerColumn = 0;
  int nSync = 0;
  std::vector<WideRound> rounds;     // planning tables (host only)
  std::vector<WideRec> recs;
  std::vector<double> recs2;         // wantW2 (the parts of a machine): second weight of every record (0.0: a one-transition candidate)
  bool wantW2 = false;
  std::vector<uint32_t> dsts;
  std::vector<WideRec> segA, segB;   // linearised streams
  int NV = 0, NX = 0;
  bool fastIdx = false;              // records carry 16-bit vector indices for both column parities
  WideRec *d_segA = nullptr, *d_segB = nullptr;
  WideDev dev{};
  bool f32 = false;                  // log-sum-exp program compiled for the single-precision relative kernel
  bool hyb = false;                  // ... with the current column in LDS and the previous one in L2
  WideRec32 *d_seg32A = nullptr, *d_seg32B = nullptr;
  unsigned long long *d_flags = nullptr;
  WideDev32 dev32{};
  // the phase-structured streams of k_wide_viterbi (built for max programs whose vectors sit in LDS with 16-bit indices)
  bool vitOk = false;
  WideRec *d_vit[4] = {nullptr, nullptr, nullptr, nullptr};
  WidePhase *d_phase = nullptr;
  WideVitDev vit{};
  // the retimed program (wide_ret_build): preferred for max programs when it fits
  bool retOk = false, retGv = false;                    // retGv: the ring lives in an L2-resident scratch vector
  WideRec *d_ret = nullptr;
  WideRetDev ret{};
  size_t retLdsBytes = 0;
  long long retW2Offset = 0;         // a part with two-transition candidates: byte offset of the second weights behind the records
  int retPeriod = 0, retTauMax = 0, retPeriodMin = 0;
  // Viterbi with ONE TRACEBACK CODE per cell instead of the fp64 cell (round 4; requested with tbCodes before wide_build): the
  // retimed max sweep also keeps, per state, the PLACE of its first maximal candidate in the reference's enumeration order
  // (emitting transitions, then silent ones, each in `incoming` order; src/dpmatrix.defs.h:93-103) and stores that byte;
  // tbEntry[tbOff[state] + code] = position in the incoming view << 16 | emitting << 15 | source state (0xFFFFFFFF: the seed)
  bool tbCodes = false, tbOk = false;
  int *d_tbOff = nullptr; uint32_t *d_tbEntry = nullptr;
  std::vector<int> h_tbOff; std::vector<uint32_t> h_tbEntry;      // host copies (the debug dump)
  long long tbEntries = 0;
  int tbFromSet = -1;                // the last traceback-code fill ran through partSets[tbFromSet] (its codes decode with that set's tables)
  bool shapeChosen = false;          // the column-by-column program was built (its closure shape is kept across weight refreshes)
  bool partsOff = false;             // latched: a partitioned launch of this program waited in vain once (wide_parts_failed) -- one workgroup per sequence until it is rebuilt
  std::vector<WideRec> h_ret;        // host copy of the retimed streams (the generated kernel's table is built from it on first use)
  WideJitKernel jit[2];              // the generated kernel of the one-workgroup sweep ([1]: fp64 correction term)
  std::vector<WidePartSet> partSets; // k workgroups per sequence: one set per k that was asked for (built on first use)
  std::vector<WidePartHint> partHints;
  size_t vecBytes32() const { return (size_t)(2 * NV + NX) * sizeof(float); }
  size_t vecBytes() const { return (size_t)(2 * NV + NX) * sizeof(double); }
};

// which machines this family takes: one-tape machines (generators: no input alphabet; recognisers: no output alphabet)
// with enough states to fill a workgroup
bool wide_applicable(const mb_machine *m);
// (re)build the program of one direction/semiring from the machine's current weights and upload it
bool wide_build(const mb_machine *m, bool backward, bool viterbi, WideProgram &P);
void wide_free(WideProgram &P);
void wide_set_accurate(bool on);      // the next retimed sum fills carry their log-sum-exp correction term in fp64 (E-step of long sequences)
// the retimed program of a machine, planned and linearised on the host only (no device): P.ret / P.retGv / P.retPeriod + the record stream
bool wide_ret_host(const mb_machine *m, bool backward, bool viterbi, WideProgram &P, std::vector<WideRec> &stream, bool tbCodes = false);
// ... and its k-part form: the record stream, geometry and tables of every part (a part's `h` holds HOST pointers into tabs[part]:
// gmap [Sloc], then impIdx [nImp]); false when the machine's graph has no cut
struct WidePartHost { WidePartDev h; std::vector<WideRec> stream; std::vector<uint32_t> tab; int period = 0, periodMin = 0; size_t ldsBytes = 0; double modelCost = 0.0; };
// W = 0: the lanes per part (and the ring depth) are searched; hint: the choice of an earlier build of the same cut (in / out)
bool wide_parts_host(const mb_machine *m, bool backward, bool viterbi, bool tbCodes, int k, int W, std::vector<WidePartHost> &parts, int &nExpTot,
                     std::vector<int> *tbOff = nullptr, std::vector<uint32_t> *tbEntry = nullptr, WidePartHint *hint = nullptr, int *Wout = nullptr, int *ringOut = nullptr);
// sweep every pair of the chunk; pool != nullptr: materialise the matrix (reference layout); loglike != nullptr: gather
// the log-likelihood of each pair.  d_desc/hp describe the same pairs (cellBase relative to pool).
// h_desc (host copy of d_desc) + cus (CUs this launch may count on): with fewer sequences than CUs the sweep runs k workgroups per
// sequence when the machine's graph can be cut (see WidePartDev); without them, or when it cannot, one workgroup per sequence
int wide_fill(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_tape, double *pool,
              double *loglike, hipStream_t st, bool lastOnly = false, const PairDesc *h_desc = nullptr, int cus = 0);
int wide_last_parts();      // k of the last launch of this family (1: one workgroup per sequence)
int wide_parts_for(const mb_machine *m, WideProgram &P, long long nPairs, int cus, const PairDesc *h_desc = nullptr);      // k such a launch would use (builds the parts); h_desc: the launch's sequences (short ones are not worth a cut)
double wide_parts_cost(const mb_machine *m, WideProgram &P, long long nPairs, int cus, const PairDesc *h_desc = nullptr);      // modelled time per column of that cut (0: none)
bool wide_parts_failed();   // after the streams were synchronised: a bounded wait ran out (error set, flag cleared, the programs latched to one workgroup per sequence)
bool wide_parts_retry();    // the API call that just failed did so because of that: run it once more (asked once)
void wide_parts_reset();    // an API call left through an error: no raised status word, no pending flag for the next one
const char *wide_kernel_name(const WideProgram &P);       // the kernel wide_fill launches for this program
// ViterbiMatrix::fill keeping one traceback code per cell (P.tbOk): tb = bytes, wide_tb_stride(S) per column, PairDesc::cellBase =
// BYTE offset of the sequence's first column; scores of the end cells in loglike
inline int wide_tb_stride(int S) { return (S + 3) & ~3; }
int wide_fill_tb(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_tape, unsigned char *tb,
                 double *loglike, hipStream_t st, const PairDesc *h_desc = nullptr, int cus = 0);
// ... and DPMatrix::traceBack over those codes: one workgroup per sequence, its first lane walks a window of code rows in LDS that
// the other wavefronts refill ahead of it (and the decode table, when it fits beside the window)
int wide_traceback_codes(const mb_machine *m, const WideProgram &P, const PairDesc *d_pairs, long long nPairs, const unsigned char *tb,
                         const double *d_loglike, const long long *d_slotOff, uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st);
// Two sweeps (e.g. Forward over one set of pairs, Backward over another) in ONE launch: workgroups 0..nA-1 run program A,
// the rest program B, so both are on the chip together whatever the queue scheduler does with two streams.  Returns -1
// (nothing launched) when the two programs do not share a kernel variant.
int wide_fill2(const mb_machine *m, WideProgram &A, WideProgram &B, const PairDesc *d_descA, const PairDesc *d_descB, long long nA, long long nB,
               const int *d_tape, double *poolA, double *poolB, hipStream_t st, bool lastOnly);
// log-likelihood of a sequence cut behind its prefix (d_pairs: the PREFIXES' descriptors): Forward column of the prefix x emitting transitions
// labelled with the token at the cut x Backward column of the suffix behind it (every path crosses the cut exactly once)
int wide_join(const mb_machine *m, const PairDesc *d_pairs, long long nPairs, const int *d_tape, const double *fvec, const double *bvec,
              double *loglike, hipStream_t st);


// posterior counts of a one-tape machine from its two materialised matrices (any state count): a lane owns a transition
struct WideCountPlan { bool ok = false; int nWaves = 0; void *d_edges = nullptr; void *d_waveLabel = nullptr; };
bool wide_counts_build(const mb_machine *m, WideCountPlan &C);
void wide_counts_free(WideCountPlan &C);
int wide_counts(const mb_machine *m, const WideCountPlan &C, const PairDesc *d_desc, const std::vector<PairDesc> &hp, const int *d_tape,
                const double *fwd, const double *bwd, double *d_counts, hipStream_t st);

// Viterbi traceback of a one-tape machine (src/dpmatrix.defs.h:82-110 on a lattice of one column per symbol): one workgroup
// per sequence, the walk by its first wavefront, the two columns a step can read in LDS (the others prefetch the next one)
struct WideTbPlan { bool tried = false, ok = false; void *d_edges = nullptr; int *d_begin = nullptr; size_t ldsBytes = 0; };
bool wide_traceback_build(const mb_machine *m, WideTbPlan &T);      // false: the machine does not fit (the generic walker takes it)
void wide_traceback_free(WideTbPlan &T);
int wide_traceback(const mb_machine *m, const WideTbPlan &T, const PairDesc *d_pairs, long long nPairs, const int *d_tape, const double *d_pool,
                   const long long *d_slotOff, uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st);

}  // namespace mb

// mb_medium.h -- host/device structures of the "lanes = states" tiled kernel family (see mb_medium.hip).
#pragma once
#include <algorithm>
#include <vector>

#include "mb_internal.h"

namespace mb {

constexpr int MED_MAXSLOT = 4;      // candidate slots evaluated together (two-pass max / sum-exp in registers)
constexpr int MED_DESC_WORDS = 8;   // descriptor words per chunk (32 B, fetched with scalar loads)
constexpr int MED_MODE_COUNT = 3;   // internal kernel mode: Forward fill (sum) fused with posterior transition counts
constexpr int MED_MODE_TB = 4;      // internal kernel mode: Viterbi fill (max) that keeps ONE traceback byte per cell instead of the fp64 cell
// what a tile kernel keeps in HBM (JMAT of the specialised kernel): MED_MAT_NONE = one workgroup sweeps a whole strip, halo
// columns only; MED_MAT_FULL = tiles + the fp64 matrix; MED_MAT_ROLL = tiles without a matrix (halo columns of the few
// states other strips read + a boundary record per strip): traceback-byte Viterbi, count sweep, log-likelihood-only Forward
enum { MED_MAT_NONE = 0, MED_MAT_FULL = 1, MED_MAT_ROLL = 2 };
constexpr int MED_GEOM_LEVELS = 5;  // strip widths a program is specialised for: its widest, halved 0..4 times

// One candidate of one lane: 16 bytes, fetched with a single global_load_dwordx4.
struct alignas(16) MedRec {
  double w;          // log-weight (padding: -inf)
  uint32_t srcOff;   // byte offset of the source value inside an LDS state vector (padding: the -inf sentinel)
  uint32_t dstOff;   // slot 0 of a chunk only: byte offset of the state this lane finalises (idle lane: the dummy entry)
};

// The compiled "program" of a machine for one sweep direction.
//   round  = up to LPG states, one per lane of a lane group, that may be finalised together;
//   slot   = one candidate (source value, log-weight) per lane; the vector a slot reads is wave-uniform:
//            0 diag (i-1,o-1), 1 left (i-1,o), 2 down (i,o-1), 3 cur (same supercell, earlier rounds);
//   chunk  = up to MED_MAXSLOT slots of one round that read the same vector, the unit the kernel software-pipelines.
// desc[chunk*MED_DESC_WORDS]: word 0 = ns | first<<4 | last<<5 | sync<<6 | nsNext<<8 (15 = none) | single<<12;
//                             words 1..4 = offset, mulI, mulO | vector<<24, stride between the chunk's slots.
// record index of (slot k, lane) = offset + inTok*mulI + outTok*mulO + laneInGroup + k*stride.
//
// Two program kinds:
//   EXACT   : silent transitions level by level, one candidate per edge in the reference's order -> every candidate
//             is the single rounded add cell+logW of the reference, Viterbi is bit-exact.
//   CLOSURE : (sum semiring only) the silent sub-graph is replaced by its transitive closure over the "base" states
//             (states fed by emitting edges, plus the start state), so a supercell needs two stages instead of one per
//             silent level: v = W* e  with  e = emit-only parts (stage 1)  and  W* = sum over silent paths (stage 2).
struct MedProgDev {
  int S, Spad, LPG, G, NS, nChunks;
  int nIn, nOut, startNode, endNode;
  unsigned seedOff;         // byte offset of the value that receives the seed 0 at the origin cell
  int ldsImageRecs;         // run-time specialised kernel: number of token-independent records kept in LDS
  const int *desc;
  const MedRec *rec;
  const MedRec *ldsImage;   // those records, in the order the specialised kernel addresses them
  const int *accMap;        // count programs with a compact loop-time accumulator table: entry -> transition (MedProgram::accMap)
};

// structure of the program, kept for the run-time code generator (mb_medium_jit.cpp)
// placement of a slot's records in the run-time specialised kernel (mb_medium_jit.cpp, medium_jit_plan):
//   REG    loop-invariant per lane (token-independent, or dependent on the column's input token only): loaded once
//          into VGPRs before the sweep;   LDS  copied into the LDS image next to the ring;   GLOBAL  fetched per step.
enum { MED_PLACE_GLOBAL = 0, MED_PLACE_LDS = 1, MED_PLACE_REG = 2 };
struct MedSlotInfo { int T; long long recBase; int place = MED_PLACE_GLOBAL; long long ldsOff = 0; };   // table (= vector it reads, = token kind), first record, placement, record offset in the LDS image
struct MedRoundInfo { std::vector<MedSlotInfo> slots; bool sync = false, single = false;
  bool flat = false;   // count programs: the usage pass behind the fill's rounds -- every record is ONE transition with its own source, destination and accumulator
  bool fused = false;  // count programs (MedProgram::fusedEmit): the round's emitting candidates (tables 0-2) are real transitions whose usage term is added right here
};

struct MedJit {                 // one specialised kernel (per program and semiring)
  bool tried = false;
  void *module = nullptr, *func = nullptr;
  size_t ldsBytes = 0;
};

struct MedProgram {
  int G = 0, LPG = 0, NS = 0, Spad = 0, nChunks = 0, nRounds = 0;
  bool backward = false, closure = false;
  bool hasSplits = false;           // high-degree states were cut into parts + combining nodes (build_program)
  bool counting = false;            // Forward fill + posterior counts program.  Levelled form (MB_MEDIUM_COUNT_FLAT=0): the exact program, the
                                    // upper 16 bits of a record's srcOff hold the byte offset of its transition's accumulator in the LDS count
                                    // array.  Flat form (default): the closure Forward program + one `flat` round of usage records
                                    // {w, srcOff = source | destination << 16, dstOff = accumulator offset}, one per transition
  bool flatCount = false;
  // Round 5 (flat form): the EMITTING transitions' usage is FUSED into the fill's emit rounds -- v = F(src) + w is in a register there and
  // every candidate of a lane shares one destination, so the term costs one B look-up per round instead of a record + two look-ups
  // per transition, and the usage pass keeps the silent transitions only (the 482-state machine: 4 output-token slots = 20 KB of LDS
  // records less).  A fused record carries its accumulator offset in the upper half of srcOff, and slot 0 of a lane the byte offset
  // of its real destination state in the Backward supercell in the upper half of dstOff.  Not when emitting candidate lists are split.
  bool fusedEmit = false;
  // ... and the accumulators are TWO tables: a compact one that lives through the step loop, for the transitions whose usage records
  // are not loop-invariant (token-selected ones: ds_add per step), and one entry per transition that exists only AFTER the loop, laid
  // over the ring (dead by then), which takes the register sums of the loop-invariant records.  The 482-state machine: 25 KB -> 8 KB.
  std::vector<int> accMap;          // loop-time table: entry -> transition
  int *d_accMap = nullptr;
  int accAllEntries = 0;            // flat count programs: nTrans + LPG entries of the after-the-loop table (0: one table, accEntries)
  int accEntries = 0;               // counting: entries of the loop-time accumulator table + LPG dummies (padding candidates, one per lane of a group); levelled form: nTrans + LPG
  std::vector<int> desc;
  std::vector<MedRec> rec;
  std::vector<int> wref;            // per record: >= 0 global edge id, -1 padding (-inf), <= -2 closure pair -2-id
  // closure structure (recomputed numerically whenever the weights change)
  std::vector<std::vector<std::pair<int, uint32_t>>> silPred;   // node -> (predecessor node, edge id)
  std::vector<char> isBase;
  std::vector<int> stageOf;                                     // closure programs: stage that finalises the state (0 = emit-only)
  std::vector<std::vector<int>> closBase;                       // node -> sorted base ancestors
  std::vector<std::vector<int>> closPair;                       // node -> pair id per ancestor (parallel to closBase)
  int nPairs = 0;
  uint32_t dummyOff = 0;            // byte offset of the write-only dummy entry of an LDS state vector
  std::vector<MedRoundInfo> roundInfo;
  std::vector<long long> ldsImageIdx;   // record indices copied into the LDS image (slots placed in LDS by medium_jit_plan)
  int regUsed = 0;                      // what the last plan spent of it
  int regBudget = -1;                   // VGPRs medium_jit_plan may spend on loop-invariant records (-1: default)
  int planC = 0, planHalo = 0, planWaves = 0;   // geometry the placement is made for (the program's own: its widest strips)
  int tokWindow = 64;                   // steps per output-token window kept in LDS by the specialised kernel
  std::vector<int> haloStates;          // states whose values another strip reads (sources of input-consuming candidates), ascending
  // IN-PLACE RING (round 5).  What a step reads from EARLIER steps is (a) its own column's cells of the step before -- the sources of
  // output-token candidates, which in a closure program all sit in the emit rounds of stage 0 -- and (b) the left column's cells of
  // one / two steps before, but only the sources of input-consuming candidates (haloStates: 3 of psw2dna's 271 states).  So the
  // matrix-free sum kernels keep per column ONE full vector, read as "the step before" by the emit rounds -- every load of stage 0 is
  // issued before its first store -- and overwritten in place by this step's cells, + NS SHORT vectors of the halo states (this step's
  // copy being written, the NS - 1 before it read by the right neighbour): 2.3 instead of 4.4 KB per column for psw2dna, 12 instead of
  // 8 wavefronts per CU.  recC = rec with the input-consuming tables' source offsets renamed into the short vector (padding: its own
  // -inf entry) and, in the upper half of slot 0's dstOff, 8 x (place of the destination in the short vector + 1) or 0.
  std::vector<unsigned char> recT;      // table of each record
  std::vector<MedRec> recC;
  MedRec *d_recC = nullptr, *d_ldsImageC = nullptr;
  bool inPlaceOk = false;               // every emit round precedes the first synchronisation point, none in the running form, few enough slots
  signed char compactState[24 * 2] = {0};   // per kernel kind (medium_jit_slot / MED_GEOM_LEVELS): 0 untried, 1 compact ring in use, -1 not
  int compactWaves[24 * 2] = {0};
  int *d_desc = nullptr;
  MedRec *d_rec = nullptr, *d_ldsImage = nullptr;
  MedProgDev dev{};
  // [(3 * medium_jit_index(mode) + matKind) * 2 + env][MedGeom::level]: sum / max / count / traceback bytes, MED_MAT_*, and the strip
  // width (level h = the program's widest strip halved h times; narrow strips for short input sequences)
  MedJit jit[24 * MED_GEOM_LEVELS];       // ... x restricted envelopes (MedGeom::env)
};

// haloSteps > 0: the materialised kernel loads the halo supercells of a whole tile (at most haloSteps steps) into LDS in
// its prologue, so its step loop issues NO vector-memory load (machines with few states, where a step is shorter than
// the time the previous step's stores need to be acknowledged)
struct MedGeom { int waves = 0, C = 0; size_t ldsBytes = 0; int haloSteps = 0; int level = 0; bool env = false;   // env: the kernel variant that clips to the pairs' envelopes
  bool compact = false;   // the IN-PLACE RING of the matrix-free sum kernels (round 5, MedProgram::inPlaceOk): more columns in the same LDS
};

// restricted envelopes of the pairs (PairDesc::envBase rows of the batch's inStart / inEnd arrays): device copies for the
// kernel, host copies for the tile lists (tiles outside every envelope row are not launched); all null = full envelopes
struct MedEnv { const int *d_start = nullptr, *d_end = nullptr, *h_start = nullptr, *h_end = nullptr; };

// host-only part (program, geometry, placement plan, numeric weights): needs no device, used by mb_debug_jit_source
bool medium_build_host(const mb_machine *m, bool backward, int closure, int G, MedProgram &P, MedGeom &geo);   // closure: 0 levelled, K >= 1 closure in K stages
bool medium_build(const mb_machine *m, bool backward, int closure, int G, MedProgram &P, MedGeom &geo);
bool medium_build_unsplit(const mb_machine *m, int G, MedProgram &P, MedGeom &geo);
// Forward program whose sweep also accumulates posterior transition usage (see MedProgram::counting); closure / cuts: the staged
// silent closure of its fill rounds (flat form only; what fast_state chose for the Forward program)
bool medium_build_count_host(const mb_machine *m, int G, MedProgram &P, MedGeom &geo, int closure = 0, const std::vector<int> &cuts = std::vector<int>());
bool medium_build_count(const mb_machine *m, int G, MedProgram &P, MedGeom &geo, int closure = 0, const std::vector<int> &cuts = std::vector<int>());
// Forward fill of the chunk's matrices fused with MachineCounts accumulation (Backward matrices given); needs the
// run-time specialised kernel: returns -1 (nothing launched) when it is unavailable, 0 ok, 1 error
int medium_counts_materialised(const mb_machine *m, MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs,
                               const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_fwd,
                               const double *d_bwd, double *d_counts, double *d_loglike, hipStream_t st, const MedEnv &env = MedEnv());
int medium_viterbi_tb(const mb_machine *m, MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs, const std::vector<PairDesc> &pairs,
                      const int *d_in, const int *d_out, unsigned char *d_tb, double *d_loglike, hipStream_t st, const MedEnv &env = MedEnv());
int medium_forward_rolltiles(const mb_machine *m, MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs, const std::vector<PairDesc> &pairs,
                             const int *d_in, const int *d_out, double *d_loglike, hipStream_t st, const MedEnv &env = MedEnv());
int medium_counts_rolling(const mb_machine *m, MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs, const std::vector<PairDesc> &pairs,
                          const int *d_in, const int *d_out, const double *d_bwd, double *d_counts, double *d_loglike, hipStream_t st,
                          const MedEnv &env = MedEnv());
// G = columns (supercells) per wavefront, LPG = 64 / G lanes per supercell.  G = 64 is "one lane per supercell": the
// mapping for machines with a handful of states (dnapsw, protpsw: 8 states).
inline bool medium_valid_G(int G) { return G >= 1 && G <= 64 && (G & (G - 1)) == 0; }
inline int medium_default_G(int S) {   // measured with the specialised kernel: psw2dna (271 states) G=4 (8 wavefronts x 256 VGPRs) >= 2 > 1
  // protpsw.translate.dnapsw (482 states, 22 silent levels, records kept in LDS): G=2 (264 G cells/s) > 4 (192) > 1 (167);
  // psw2dna (271 states): 4 >= 2 > 1; dnapsw/protpsw (8 states): 32 > 16 > 64 > 8
  return S >= 1024 ? 1 : (S >= 384 ? 2 : (S >= 48 ? 4 : (S >= 24 ? 8 : 32)));
}
// count sweep: LDS count atomics collide across the lanes of a wavefront that share a transition (few states: 16 columns
// per wavefront, not 32); the Backward supercell per column in LDS caps the strip at 16 columns for psw2dna (271 states),
// where 8 wavefronts x 2 columns (133 ms at 64 x 487 x 2 kb) beat 4 x 4 (150) and 16 x 1 (164)
// (round 4, flat count program: 482 states -- 6 columns of LDS whatever G is -- run 1 column per wavefront: 87 vs 76 G lattice-cells/s)
inline int medium_default_count_G(int S) { return S >= 384 ? 1 : (S >= 128 ? std::min(medium_default_G(S), 2) : (S >= 24 ? medium_default_G(S) : 16)); }
inline int medium_jit_index(int mode) { return mode == MB_VITERBI ? 1 : (mode == MED_MODE_COUNT ? 2 : (mode == MED_MODE_TB ? 3 : 0)); }
void medium_eval_weights(const mb_machine *m, MedProgram &P);
bool medium_refresh_weights(const mb_machine *m, MedProgram &P);
void medium_jit_plan(const mb_machine *m, MedProgram &P, const MedGeom &geo);
void medium_count_layout(const mb_machine *m, MedProgram &P);      // flat count programs, after every medium_jit_plan: accumulator offsets of the usage records by placement
int medium_jit_max_cands();
long long medium_inplace_kernels();      // kernel kinds that took the in-place ring so far (introspection)
long long medium_jit_spill_count(const std::string &codeObject);
size_t medium_jit_lds_bytes(const MedProgram &P, const MedGeom &geo, int mode = MB_FORWARD);
std::string medium_jit_source(const mb_machine *m, const MedProgram &P, const MedGeom &geo, int mode, int matKind);
// traceback bytes of the tiled family: bytes per supercell in HBM (reference order, one byte per state, padded so that a
// lane stores 16 at a time), and whether the machine's Viterbi sweep can keep them (code = table << 6 | index)
inline int medium_tb_stride(int S) { return (S + 15) & ~15; }
bool medium_tb_eligible(const mb_machine *m, const MedProgram &P);
void medium_free(MedProgram &P);
// stage boundaries of the next closure programs built (first silent level of stages 2, 3, ...); empty = even level groups
void medium_set_cuts(const std::vector<int> &cuts);
bool medium_geometry(const mb_machine *m, const MedProgram &P, MedGeom &geo);
void medium_fit_records(const mb_machine *m, const MedProgram &P, MedGeom &geo);
int medium_fill_materialised(const mb_machine *m, MedProgram &P, const MedGeom &geo, int mode, int startNode,
                             const PairDesc *d_pairs, const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out,
                             double *d_pool, hipStream_t st, const MedEnv &env = MedEnv());
// run-time specialisation (mb_medium_jit.cpp): returns false if hiprtc is unavailable or the program does not qualify
bool medium_jit_get(const mb_machine *m, MedProgram &P, const MedGeom &geo, int mode, int matKind, bool allowReplan = true);
inline int medium_jit_slot(int mode, int matKind, int level, bool env = false) { return ((3 * medium_jit_index(mode) + matKind) * 2 + (env ? 1 : 0)) * MED_GEOM_LEVELS + level; }
inline int medium_compact_len(const MedProgram &P) { return ((int)P.haloStates.size() + 2) & ~1; }      // halo states + the -inf entry, even
// geometry of a matrix-free sweep: the compact ring's (tried once per kernel kind: its kernel must compile into the registers its
// wavefront count leaves, without touching the program's placement) or the plain one
MedGeom medium_roll_geometry(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const std::vector<PairDesc> &pairs, int mode, int matKind, bool env, bool materialiseRule);
inline bool medium_jit_ready(const MedProgram &P, int mode, int matKind) {
  for (int h = 0; h < MED_GEOM_LEVELS; ++h) if (P.jit[medium_jit_slot(mode, matKind, h, false)].func || P.jit[medium_jit_slot(mode, matKind, h, true)].func) return true;
  return false;
}
// strip width for a batch (narrower strips for short input sequences)
MedGeom medium_pick_geometry(const MedProgram &P, const MedGeom &geo, const std::vector<PairDesc> &pairs, bool materialise);
void medium_jit_free(MedProgram &P);
int medium_forward_pipelined(const mb_machine *m, MedProgram &P, const MedGeom &geo, const std::vector<PairDesc> &pairs,
                             const int *d_in, const int *d_out, double *d_pool, long long poolCells, double *d_loglike,
                             hipStream_t st);
int medium_forward_rolling(const mb_machine *m, MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs,
                           const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_colHalo,
                           const long long *d_haloBase, double *d_loglike, hipStream_t st);

}  // namespace mb

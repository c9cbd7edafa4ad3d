// mb_small.h -- the "lane = column, states in registers" kernel family for machines with a handful of states
// (dnapsw, protpsw and the like: BASELINE configs 1-3).  See mb_small.cpp for the design.
#pragma once
#include <string>
#include <vector>

#include "mb_internal.h"

namespace mb {

// kernel modes of the run-time specialised sweep
enum { SM_SUM = 0,     // log-sum-exp semiring: Forward, or Backward on the reversed program
       SM_MAX = 1,     // max semiring, fp64 cells (ViterbiMatrix as a matrix: mb_fill)
       SM_TB = 2,      // max semiring, ONE traceback byte per cell instead of the fp64 cell (mb_batch_viterbi)
       SM_COUNT = 3,   // Forward sweep that reads the Backward matrix and accumulates posterior transition counts
       SM_NMODE = 4 };

// One candidate of a state: which neighbour supercell it reads (T: 0 match = (i-1,o-1), 1 input-only = (i-1,o),
// 2 output-only = (i,o-1), 3 silent = same supercell), which state there, and which weight / edge-id table it uses.
struct SmSlot { int T, src, dup, tab; };

struct SmJit { bool tried = false; void *module = nullptr, *func = nullptr; size_t ldsBytes = 0; int vgprs = 0; bool scratch = false; };   // scratch: the kernel spills to scratch memory even at one wavefront per SIMD

struct SmallProgram {
  bool ok = false, backward = false;
  int S = 0, nIn = 0, nOut = 0;
  long long nTrans = 0;
  std::vector<int> order;                    // evaluation order of the states inside a supercell
  std::vector<std::vector<SmSlot>> cand;     // [state]: candidates in the reference's enumeration order
  int nTab[4] = {0, 0, 0, 0};                // tables per kind T
  long long off[4] = {0, 0, 0, 0};           // first entry of kind T's tables in w[] / eid[]
  long long nEntries = 0;
  // layout of w[] / eid[]: [silent: nTab[3]] [input: nTab[1] x (nIn+1)] [output: nTab[2] x (nOut+1)] [match: nTab[0] x (nIn+1)(nOut+1)]
  std::vector<double> w;
  std::vector<int> eid;                      // global edge id of (slot, tokens), -1 where the token pair has no such edge
  std::vector<int> needLeft, needDiag, needDown, saveCells;   // states whose values cross a step (sorted)
  int H = 0;                                 // doubles per halo row = needLeft.size()
  int NBD = 0;                               // doubles per lane in the tile-boundary record
  int seedState = 0, endState = 0;
  bool seedSimple = true;                    // the seed state has no silent candidates: cell = origin ? 0 : fold(candidates)
  double *d_w = nullptr;
  int *d_eid = nullptr;
  // traceback decode table (forward programs): per state its candidates as {T, src, tab}
  std::vector<int> decOff;                   // [S+1]
  std::vector<uint32_t> dec;                 // T | src << 8 | tab << 16
  int *d_decOff = nullptr;
  uint32_t *d_dec = nullptr;
  SmJit jit[SM_NMODE][2][2];                 // [mode][materialise][restricted envelopes]
};

// Matrix storage of this family (device only; mb_fill converts to the reference's layout): strip a of a pair holds its
// Te = even(outLen + 64) steps one after the other; a step is NCH chunks of CB bytes for each of the 64 lanes, chunk-major,
// so that every store instruction of a wavefront writes 64 x CB contiguous bytes.
__host__ __device__ inline int small_steps(int outLen) { return (outLen + 64 + 1) & ~1; }
__host__ __device__ inline int small_strips(int inLen) { return (inLen + 64) / 64; }
__host__ __device__ inline int small_chunk_bytes(int S) { return (S % 2 == 0) ? 16 : 8; }
__host__ __device__ inline int small_chunks(int S) { return S * 8 / small_chunk_bytes(S); }
__host__ __device__ inline long long small_pair_doubles(int S, int inLen, int outLen) { return (long long)small_strips(inLen) * small_steps(outLen) * 64 * S; }
__host__ __device__ inline int small_tb_stride(int S) { return 4 * ((S + 3) / 4); }
__host__ __device__ inline long long small_pair_tb_bytes(int S, int inLen, int outLen) { return (long long)small_strips(inLen) * small_steps(outLen) * 64 * small_tb_stride(S); }

bool small_eligible(const mb_machine *m);
bool small_build_host(const mb_machine *m, bool backward, SmallProgram &P);
bool small_build(const mb_machine *m, bool backward, SmallProgram &P);
bool small_refresh_weights(const mb_machine *m, SmallProgram &P);
void small_free(SmallProgram &P);
std::string small_jit_source(const SmallProgram &P, int mode, bool materialise, bool env = false, int minWaves = 0);
int small_default_minwaves(int mode, bool env);
size_t small_jit_lds_bytes(const SmallProgram &P, int mode);
bool small_jit_get(SmallProgram &P, int mode, bool materialise, bool env = false);
bool small_can_run(SmallProgram &P, int mode, bool materialise, bool env);   // built, loaded and free of scratch memory
bool small_count_fits(SmallProgram &P, bool env);
const char *small_kernel_name(const SmallProgram &P, int mode, bool materialise);

// Per-pair placement of the buffers a sweep uses (all offsets relative to the chunk's workspaces)
struct SmAux { long long pool, halo, bound, tb; };

// One sweep over a set of pairs (device arrays already placed): launches the wavefront of tiles.
struct SmSweep {
  const PairDesc *d_pairs = nullptr;           // device copy of `pairs`
  const std::vector<PairDesc> *pairs = nullptr;
  const int *d_in = nullptr, *d_out = nullptr;
  const SmAux *d_aux = nullptr;
  double *d_pool = nullptr;                    // matrices written (materialise) or read (count mode: the Backward matrices)
  unsigned char *d_tb = nullptr;
  double *d_halo = nullptr, *d_bound = nullptr;
  double *d_loglike = nullptr;                 // [nPairs]
  const double *d_bwdLL = nullptr;             // count mode
  double *d_counts = nullptr; int nRep = 0;    // count mode: nRep replicas of [nTrans]
  SmTileCache *tileCacheFwd = nullptr, *tileCacheBwd = nullptr;   // tile lists of this set of pairs by sweep direction, kept on the device between calls (may be null)
  const int *d_envStart = nullptr, *d_envEnd = nullptr;   // restricted envelopes (Envelope::inStart / inEnd rows at PairDesc::envBase); null: all full
  const int *h_envStart = nullptr, *h_envEnd = nullptr;   // their host copies: tiles outside every envelope row are not launched
  long long haloDoubles = 0;                   // size of d_halo (pre-filled with -inf when tiles are skipped)
};
int small_sweep(SmallProgram &P, int mode, bool materialise, const SmSweep &sw, hipStream_t st);

// kernels compiled ahead of time (mb_small_kernels.hip)
int launch_small_unpack(const double *d_pool, int S, int inLen, int outLen, bool reversed, double *d_cells, const int *d_envStart,
                        const int *d_envEnd, hipStream_t st);
int launch_fill_neg_inf(double *d, long long n, hipStream_t st);
int launch_small_traceback(const SmallProgram &P, const PairDesc *d_pairs, long long nPairs, const int *d_in, const int *d_out,
                           const unsigned char *d_tb, const SmAux *d_aux, const double *d_ll, const long long *d_slotOff,
                           uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st);

}  // namespace mb

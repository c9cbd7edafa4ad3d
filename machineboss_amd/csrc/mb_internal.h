// mb_internal.h -- structures shared by the host-side machine compiler and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "mbhip.h"

namespace mb {

// ---- device view of a flattened machine (all pointers are device memory) --------------------------------
// Label key of an edge: key(inTok,outTok) = inTok*(nOut+1) + outTok over the (nIn+1)x(nOut+1) grid incl. epsilon.
// CSR row = state*K + key.  Edge arrays are permuted into the reference's iteration order (see mbhip.h).
struct DevMachine {
  int S, nIn, nOut, K;
  int nLevF, nLevB;          // number of silent levels, forward / backward direction
  // incoming view (Forward, Viterbi, traceback):   src/eval.h:66-68 `incoming`
  const int *inOff;          // [S*K + 1]
  const uint32_t *inSrc;     // [nTrans] source state
  const double *inW;         // [nTrans] log weight
  const uint32_t *inEid;     // [nTrans] global edge id
  // outgoing view (Backward, counts):               src/eval.h:66-68 `outgoing`
  const int *outOff;         // [S*K + 1]
  const uint32_t *outDst;    // [nTrans]
  const double *outW;        // [nTrans]
  const uint32_t *outEid;    // [nTrans]
  const uint16_t *eInTok, *eOutTok;  // [nTrans] labels by global edge id (traceback steps)
  // states grouped by silent level
  const int *levFOff, *levFState;    // [nLevF+1], [S]
  const int *levBOff, *levBState;    // [nLevB+1], [S]
};

// One sequence pair of a batch.
struct PairDesc {
  long long inBase, outBase;   // offsets into the batch token arrays
  int inLen, outLen;
  long long cellBase;          // offset (in doubles) of this pair's matrix inside a matrix pool
  int launch0;                 // tiled kernels: index of the launch in which this pair's first tile runs
  int pad;
  long long envBase;           // offset of this pair's envelope rows in the batch's inStart/inEnd arrays, -1 = full envelope
};

}  // namespace mb

// ---- host objects behind the opaque C handles --------------------------------------------------------------
struct mb_machine {
  int S = 0, nIn = 0, nOut = 0, K = 0;
  long long nTrans = 0;
  // host copies (global edge id order)
  std::vector<uint32_t> src, dst;
  std::vector<uint16_t> inTok, outTok;
  std::vector<double> logW;
  // derived on host
  std::vector<int> inOff, outOff;
  std::vector<uint32_t> inPerm, outPerm;   // CSR position -> global edge id
  std::vector<int> levF, levB;             // per-state level
  std::vector<int> levFOff, levFState, levBOff, levBState;
  int nLevF = 0, nLevB = 0;
  bool hasMatch = false, hasIns = false, hasDel = false;
  int maxInDeg = 0;
  // device arrays
  int *d_inOff = nullptr, *d_outOff = nullptr;
  uint32_t *d_inSrc = nullptr, *d_inEid = nullptr, *d_outDst = nullptr, *d_outEid = nullptr;
  double *d_inW = nullptr, *d_outW = nullptr;
  uint16_t *d_eInTok = nullptr, *d_eOutTok = nullptr;
  int *d_levFOff = nullptr, *d_levFState = nullptr, *d_levBOff = nullptr, *d_levBState = nullptr;
  mb::DevMachine dev{};
  void *fast = nullptr;   // kernel-family specific compiled tables (owned; see mb_fast_*.hip)
};

namespace mb {
// tile lists of one chunk of a batch for the small-machine family (mb_small.cpp), kept on the device between calls
// (d_deps / d_flags: the one-launch form of a sweep -- per tile the list positions of the tiles it reads from, and a "done" word per tile)
struct SmTileCache { long long p0 = -1, p1 = -1; int TS = 0; long long envVersion = -1; void *d_tiles = nullptr; std::vector<long long> off; void *d_deps = nullptr; void *d_flags = nullptr; };
}

struct mb_batch {
  mb_machine *m = nullptr;
  long long nPairs = 0;
  std::vector<mb::PairDesc> pairs;   // host copy, cellBase for a fully materialised pool
  long long totalCells = 0;          // sum (inLen+1)(outLen+1)S
  long long maxPairCells = 0;
  int *d_in = nullptr, *d_out = nullptr;
  mb::PairDesc *d_pairs = nullptr;
  long long nInTokTotal = 0, nOutTokTotal = 0;
  // envelopes (src/seqpair.h:75-97): cell (x,y) of pair p exists <=> envStart[envBase+y] <= x < envEnd[envBase+y]
  bool hasEnv = false;
  int *d_envStart = nullptr, *d_envEnd = nullptr;
  std::vector<int> h_envStart, h_envEnd;   // host copies: the tiled families skip tiles that lie outside every envelope row
  long long envVersion = 0;                // bumped by mb_batch_set_envelopes (cached tile lists depend on it)
  std::vector<mb::SmTileCache> smTiles;   // by 2 * chunk number + (backward sweep)
};

namespace mb {
void set_error(const std::string &msg);
// Tuning knobs: `mb_set_option` keeps them in a table of the LIBRARY -- it does not write the process environment --
// and every reader asks here: the table first, then the environment (the defaults a host sets before it starts).
const char *opt_env(const char *name);
void opt_set(const char *name, const char *value);      // value nullptr / "": back to the environment's (or the built-in) default
// MB_DETERMINISTIC=1 (read when a count call begins): posterior counts are summed in 64-bit FIXED POINT wherever the order of the
// additions depends on scheduling (LDS accumulators shared by wavefronts, global accumulators shared by tiles) -- integer addition
// is associative, so `--counts / --train` reproduce bit for bit from run to run like the reference's serial loop
// (src/counts.cpp:37-64).  Scales: 2^-44 inside a tile (a tile's partial sum stays below 2^11), 2^-36 in global memory (a count
// below 6.7e7 per call; beyond it the call fails, det_to_double); lane-private partial sums (registers, fixed shuffle trees) are deterministic as they are.
extern bool g_deterministic;
constexpr double MB_DET_TILE_SCALE = 17592186044416.0;   // 2^44
constexpr double MB_DET_GLOBAL_SCALE = 68719476736.0;     // 2^36
// A global fixed-point accumulator as a count.  The device clamps every term to [0, 2^62] before its cast (a NaN, a negative or a
// huge value saturates), so an accumulator that reaches 2^62 -- a count beyond 6.7e7, or a saturated term -- says "out of range":
// the call then FAILS instead of returning wrapped garbage with rc = 0.
inline bool det_to_double(unsigned long long u, double &out) { out = (double)u / MB_DET_GLOBAL_SCALE; return u < (1ull << 62); }
bool hip_ok(hipError_t e, const char *what);
extern hipStream_t g_stream;
extern thread_local long long g_last_launches;   // kernel launches of the dominant kernel in the last batch call
extern int g_kernel_choice;
extern size_t g_mem_budget;
// cached small device buffers (mb_api.hip): per-call descriptor / offset / tile-list arrays
hipError_t sm_alloc(void **out, size_t bytes);
void sm_free(void *p);
// grow-only device workspaces by slot (mb_api.hip); pinned for the duration of the current API call
void *ws_get(int slot, size_t bytes);
size_t budget_bytes();
// host -> device copy of a large pageable buffer through pinned staging buffers (a direct hipMemcpy pins fresh pageable
// pages on the fly, which sporadically costs tens of milliseconds)
int h2d_large(void *dstDev, const void *src, size_t bytes);
int launch_fill_neg_inf(double *d, long long n, hipStream_t st);
#define MB_HIP(call) do { if (!mb::hip_ok((call), #call)) return 1; } while (0)

// host-side machine compiler (mb_machine.cpp)
bool compile_machine(mb_machine *m, std::string *err);
bool upload_machine(mb_machine *m);
bool upload_weights(mb_machine *m);
void free_machine_device(mb_machine *m);
}  // namespace mb

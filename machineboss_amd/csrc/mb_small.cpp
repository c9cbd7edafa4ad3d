// mb_small.cpp -- "lane = column, states in registers": the kernel family for machines with a handful of states.
//
// BASELINE configs 1-3 run on 8-state machines (dnapsw, protpsw).  The tiled family (mb_medium.hip) spends a supercell's
// states on the lanes of a lane group and keeps the anti-diagonals in an LDS ring, which for 8 states means two lanes per
// supercell, four rounds with a wave-level LDS round trip each, and a workgroup barrier per step.  Here instead
//   * a WAVEFRONT is a strip of 64 consecutive input positions, one lane per column, sweeping the output axis with the
//     usual skew (lane c works on output position t - c at step t); the states of a lane's supercell are named scalars
//     in VGPRs, evaluated by straight-line code in state order -- the reference's own order (src/forward.defs.h:32-43),
//     so silent transitions need no synchronisation at all;
//   * the three neighbour supercells come from registers: (i,o-1) is the lane's own previous result, (i-1,o) is the left
//     lane's previous result (ONE v_mov_b32_dpp wave_shr:1 per 32 bits -- only for the states that are sources of
//     input-consuming transitions), (i-1,o-1) is what that shift delivered a step earlier.  Lane 0 takes its left
//     neighbour from the halo column the previous strip wrote (staged in LDS 64 steps at a time, read as a broadcast);
//   * no workgroup barrier, no LDS ring: the four wavefronts of a workgroup are four independent tiles that only share
//     the weight tables (output-token and match tables in LDS; input-token weights are per-lane registers loaded once
//     per tile; silent weights are scalar registers);
//   * a strip is cut into tiles of TS steps; tile (strip a, block b) runs in launch 2a + b, so kernel boundaries are
//     the only synchronisation between tiles (same scheme as the tiled family); the register state a tile hands to the
//     next block of its strip goes through a small boundary record, the last column through the halo column;
//   * matrices are stored TILE-MAJOR (strip, step, chunk, lane): every store instruction of a wavefront writes 64 x 16
//     contiguous bytes.  In the reference's layout ((outPos * (inLen+1) + inPos) * nStates) the 64 supercells of a step
//     are 64 pieces of 64 bytes, (inLen) supercells apart: measured 3.7 TB/s against 6.3 TB/s for a streaming fill
//     (scripts/micro/store_pattern_probe.hip).  mb_fill() converts to the reference's layout on the way out;
//   * Viterbi stores ONE traceback byte per cell (the index of the first maximal candidate in the reference's
//     enumeration order, src/dpmatrix.defs.h:93-103) instead of the fp64 cell -- SURVEY.md section 8(d): 1 B per cell;
//   * the posterior-count sweep (src/backward.cpp:58-87) is a Forward sweep that reads the Backward matrix (16 B per
//     lattice cell moved in total: Backward written once, read once; Forward never touches HBM) and keeps usage sums in
//     registers (silent and input-token transitions: the transition is fixed for a lane's whole tile) or in lane-private
//     LDS rows indexed by the output token (no two lanes ever share an address), folded into a workgroup table and
//     flushed with one fp64 atomic per used transition per four tiles.
//
// Arithmetic: as in the tiled family -- candidates, maxima and cells are fp64; the log-sum-exp correction term is
// evaluated in fp32 (v_exp_f32 / v_log_f32); max mode is exact, so Viterbi scores and traceback bytes reproduce the
// reference's choices bit for bit.
#include "mb_small.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <sstream>

#include "mb_jit.h"

namespace mb {

static int env_int_s(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}

static const int SM_MAX_STATES = 16;

bool small_eligible(const mb_machine *m) {
  if (m->S < 1 || m->S > std::min(env_int_s("MB_SMALL_MAX_STATES", SM_MAX_STATES), SM_MAX_STATES)) return false;   // (the knob can only lower the limit: register tables, traceback words and decode entries are sized for 16)
  if (m->nIn < 1 || m->nOut < 1) return false;            // one-tape machines: other families
  if (m->nTrans > 8000) return false;                     // the workgroup's count table lives in LDS
  for (long long e = 0; e < m->nTrans; ++e)
    if (m->inTok[e] == 0 && m->outTok[e] == 0 && m->dst[e] <= m->src[e]) return false;   // silent self-loop on state 0
  return true;
}

// ---- program: candidates per state in the reference's enumeration order ------------------------------------------------
bool small_build_host(const mb_machine *m, bool backward, SmallProgram &P) {
  P = SmallProgram();
  if (!small_eligible(m)) return false;
  const int S = m->S, nIn = m->nIn, nOut = m->nOut;
  P.backward = backward; P.S = S; P.nIn = nIn; P.nOut = nOut; P.nTrans = m->nTrans;
  const std::vector<int> &off = backward ? m->outOff : m->inOff;
  const std::vector<uint32_t> &perm = backward ? m->outPerm : m->inPerm;
  auto other = [&](uint32_t e) { return (int)(backward ? m->dst[e] : m->src[e]); };
  auto row = [&](int st, int it, int ot) { return ((long long)st * (nIn + 1) + it) * (nOut + 1) + ot; };
  P.order.resize(S);
  for (int k = 0; k < S; ++k) P.order[k] = backward ? S - 1 - k : k;
  P.seedState = backward ? S - 1 : 0;
  P.endState = backward ? 0 : S - 1;
  P.cand.assign(S, {});
  // slot (state, T, src, dup) -> per-token (weight, edge)
  struct Tab { int T; std::vector<int> e; };   // e: edge id per token index (-1: none)
  std::vector<Tab> tabs[4];
  for (int d = 0; d < S; ++d) {
    for (int T = 0; T < 4; ++T) {
      // token grid of this kind
      const int nI = (T == 0 || T == 1) ? nIn : 0, nO = (T == 0 || T == 2) ? nOut : 0;
      const int nTok = T == 0 ? (nIn + 1) * (nOut + 1) : (T == 1 ? nIn + 1 : (T == 2 ? nOut + 1 : 1));
      std::map<std::pair<int, int>, std::vector<int>> slots;   // (src, dup) -> edge per token index
      auto visit = [&](int it, int ot, int tokIdx) {
        const long long rw = row(d, it, ot);
        int prevSrc = -1, dup = 0;
        for (int a = off[rw]; a < off[rw + 1]; ++a) {
          const uint32_t e = perm[a];
          const int o = other(e);
          if (T == 3 && (backward ? o <= d : o >= d)) return false;   // excluded by small_eligible
          dup = (o == prevSrc) ? dup + 1 : 0;
          prevSrc = o;
          auto &v = slots[{o, dup}];
          if (v.empty()) v.assign(nTok, -1);
          v[tokIdx] = (int)e;
        }
        return true;
      };
      if (T == 3) { if (!visit(0, 0, 0)) return false; }
      else if (T == 1) { for (int it = 1; it <= nI; ++it) visit(it, 0, it); }
      else if (T == 2) { for (int ot = 1; ot <= nO; ++ot) visit(0, ot, ot); }
      else for (int it = 1; it <= nI; ++it) for (int ot = 1; ot <= nO; ++ot) visit(it, ot, it * (nOut + 1) + ot);
      for (auto &kv : slots) {
        P.cand[d].push_back({T, kv.first.first, kv.first.second, (int)tabs[T].size()});
        tabs[T].push_back({T, kv.second});
      }
    }
    if (P.cand[d].size() > 255) return false;
  }
  for (const SmSlot &sl : P.cand[P.seedState]) if (sl.T == 3) P.seedSimple = false;
  if (!P.seedSimple) return false;   // (a start state fed by silent transitions: the other families take it)
  for (int T = 0; T < 4; ++T) P.nTab[T] = (int)tabs[T].size();
  if (P.nTab[3] > 48 || P.nTab[1] > 48) return false;   // scalar / vector registers held for the whole tile
  const long long szT[4] = {(long long)(nIn + 1) * (nOut + 1), nIn + 1, nOut + 1, 1};
  P.off[3] = 0;
  P.off[1] = P.off[3] + P.nTab[3];
  P.off[2] = P.off[1] + (long long)P.nTab[1] * szT[1];
  P.off[0] = P.off[2] + (long long)P.nTab[2] * szT[2];
  P.nEntries = P.off[0] + (long long)P.nTab[0] * szT[0];
  if ((P.nEntries - P.off[2]) * 8 > 64 * 1024) return false;   // LDS copy of the output-token and match tables
  P.eid.assign((size_t)P.nEntries, -1);
  for (int T = 0; T < 4; ++T)
    for (int k = 0; k < P.nTab[T]; ++k)
      for (long long j = 0; j < szT[T]; ++j) P.eid[(size_t)(P.off[T] + k * szT[T] + j)] = tabs[T][k].e[(size_t)j];
  // which states cross a step
  std::set<int> L, Dg, Dn;
  for (int d = 0; d < S; ++d)
    for (const SmSlot &sl : P.cand[d]) {
      if (sl.T == 0) { Dg.insert(sl.src); L.insert(sl.src); }
      else if (sl.T == 1) L.insert(sl.src);
      else if (sl.T == 2) Dn.insert(sl.src);
    }
  P.needLeft.assign(L.begin(), L.end()); P.needDiag.assign(Dg.begin(), Dg.end()); P.needDown.assign(Dn.begin(), Dn.end());
  std::set<int> sv(L.begin(), L.end()); sv.insert(Dn.begin(), Dn.end());
  P.saveCells.assign(sv.begin(), sv.end());
  P.H = (int)P.needLeft.size();
  P.NBD = std::max(1, (int)(P.saveCells.size() + P.needDiag.size()));
  // traceback decode table
  P.decOff.assign(S + 1, 0);
  for (int d = 0; d < S; ++d) {
    P.decOff[d + 1] = P.decOff[d] + (int)P.cand[d].size();
    for (const SmSlot &sl : P.cand[d]) P.dec.push_back((uint32_t)sl.T | ((uint32_t)sl.src << 8) | ((uint32_t)sl.tab << 16));
  }
  P.w.assign((size_t)P.nEntries, -INFINITY);
  for (size_t k = 0; k < P.w.size(); ++k) if (P.eid[k] >= 0) P.w[k] = m->logW[P.eid[k]];
  P.ok = true;
  return true;
}

template <class T>
static bool up_s(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(small program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D(small program)")) return false;
  return true;
}

bool small_refresh_weights(const mb_machine *m, SmallProgram &P) {
  if (!P.ok) return true;
  for (size_t k = 0; k < P.w.size(); ++k) P.w[k] = P.eid[k] >= 0 ? m->logW[P.eid[k]] : -INFINITY;
  if (!P.d_w) return up_s(P.d_w, P.w);
  return hip_ok(hipMemcpy(P.d_w, P.w.data(), P.w.size() * sizeof(double), hipMemcpyHostToDevice), "H2D(small weights)");
}

bool small_build(const mb_machine *m, bool backward, SmallProgram &P) {
  if (!small_build_host(m, backward, P)) return false;
  if (!(up_s(P.d_w, P.w) && up_s(P.d_eid, P.eid) && up_s(P.d_decOff, P.decOff) && up_s(P.d_dec, P.dec))) { small_free(P); return false; }
  return true;
}

void small_free(SmallProgram &P) {
  for (int mo = 0; mo < SM_NMODE; ++mo)
    for (int ma = 0; ma < 2; ++ma)
      for (int en = 0; en < 2; ++en)
        if (P.jit[mo][ma][en].module) (void)hipModuleUnload((hipModule_t)P.jit[mo][ma][en].module);
  void *ptrs[] = {P.d_w, P.d_eid, P.d_decOff, P.d_dec};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  P = SmallProgram();
}

// ---- LDS budget ---------------------------------------------------------------------------------------------------------
static long long lds_w_doubles(const SmallProgram &P) { return ((P.nEntries - P.off[2]) + 1) & ~1ll; }
static int row_floats(const SmallProgram &P) {   // count mode: lane-private usage rows, one float per (match table, output token)
  int n = P.nTab[0] * (P.nOut + 1);
  return n | 1;   // odd stride: the 64 rows start in different LDS banks
}
static int outacc_floats(const SmallProgram &P) { return (P.nTab[2] * (P.nOut + 1) + 1) & ~1; }   // count mode: usage by (output-token table, token)
static long long wave_doubles(const SmallProgram &P, int mode) {
  long long d = 32 + 64 + 64ll * std::max(P.H, 1);   // tokens, envelope rows (start / end), halo rows
  if (mode == SM_COUNT) d += (64ll * row_floats(P) + (row_floats(P) & 1) + outacc_floats(P) + 1) / 2;   // rowL, then outAcc (the kernel starts it (JROWF & 1) floats further on)
  return (d + 1) & ~1ll;
}
size_t small_jit_lds_bytes(const SmallProgram &P, int mode) {
  long long d = lds_w_doubles(P) + 4 * wave_doubles(P, mode);
  if (mode == SM_COUNT) d += (P.nTrans + 1) & ~1ll;
  return (size_t)d * 8;
}

// ---- source generator ----------------------------------------------------------------------------------------------------
static const char *kSmallSkeleton = R"MBSM(
/*@DEFS@*/
#define NEG_INF (-__builtin_inf())
#define SM_L2E 1.44269504088896f
#define SM_LN2 0.693147180559945f
struct PairDesc { long long inBase, outBase; int inLen, outLen; long long cellBase; int launch0; int pad; long long envBase; };
struct SmAux { long long pool, halo, bound, tb; };
struct SmallArgs {
  const PairDesc *pairs; const int *inTok; const int *outTok;
  const int4 *tiles; int tileBase, tileEnd, TS, nRep;
  double *pool; unsigned char *tb; double *halo; double *bound; const SmAux *aux;
  double *loglike; const double *w; const int *eid; const double *bwdLL; double *counts;
  const int *envStart; const int *envEnd;
  const int4 *deps; unsigned *flags; unsigned *err; long long timeoutTicks;      // the one-launch form of a sweep (flags == nullptr: one launch per wavefront of tiles)
};
// ONE LAUNCH for a whole sweep (batches whose launches hold a handful of tiles: a single 1 kb x 1 kb pair is a chain of 47 dependent
// launches of ~20 us each): every tile of the sweep is in the grid, in wavefront order, and waits for the tiles it reads from -- the
// block before it in its strip (boundary record) and the blocks of the strip to its left whose halo rows it reads -- through one
// "done" word per tile.  A tile only ever waits for tiles EARLIER in the list, which the dispatcher starts first; the wait is bounded
// (the host then runs the sweep launch by launch).
__device__ __forceinline__ void sm_wait(const unsigned *p, unsigned *err, long long timeoutTicks) {
  const long long t0 = (long long)wall_clock64();
  while (!__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if ((long long)wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    __builtin_amdgcn_s_sleep(2);
  }
}
// PERSISTENT STRIPS (SmallArgs::TS beyond the pair's steps: a strip is ONE tile, no tile set-up per block).  The halo column is its own
// hand-over: the host fills it with a sentinel (all ones: a NaN no cell ever holds), lane 63 stores its rows with agent-scope atomic
// stores as the steps produce them -- no fence, no flag --, and the strip to the right stages the rows of its next JSUB steps with
// agent-scope loads, repeating while a row still reads as the sentinel (bounded, like sm_wait).  A strip then lags its left neighbour
// by 63 + JSUB steps (+ one load's latency) instead of two 64-step blocks.
#define HALO_EMPTY (~0ull)
__device__ __forceinline__ void halo_store(double *p, double v, bool jper) {
  if (jper) __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ double halo_wait(const double *p, unsigned *err, long long timeoutTicks) {
  unsigned long long b = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b != HALO_EMPTY) return __longlong_as_double((long long)b);
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    b = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b != HALO_EMPTY) return __longlong_as_double((long long)b);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return NEG_INF;
    if ((long long)wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return NEG_INF; }
    __builtin_amdgcn_s_sleep(1);
  }
}
typedef const __attribute__((address_space(4))) double *cdbl_t;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2a8 __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }   // operands are never NaN
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float ex2(double d) { return __builtin_amdgcn_exp2f((float)d * SM_L2E); }
// lane l takes the value of lane l-1; lane 0 takes `old`
__device__ __forceinline__ int shri(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr1(double v, double old) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float shrf(float v, float old) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// log(exp(a) + exp(b)): fp64 maximum, correction term log(1 + exp(min - max)) in fp32
__device__ __forceinline__ double lse2(double a, double b) {
  const double mx = dmax(a, b), mn = dmin(a, b);
  float df = (float)(mn - mx);                    // NaN when both are -inf
  df = (mx == NEG_INF) ? 0.0f : df;
  const float e = __builtin_amdgcn_exp2f(df * SM_L2E);
  return mx + (double)(__builtin_amdgcn_logf(1.0f + e) * SM_LN2);
}
// sum over the 64 lanes (butterfly through the LDS crossbar; every lane ends up with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ void lds_add_f32(float *p, float x) {
  (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the workgroup's count table: fp64, or -- deterministic mode (jdet in scope: bit 16 of SmallArgs::nRep, mb_internal.h
// g_deterministic) -- 64-bit fixed point at 2^-44, whose additions commute
#define lds_add_f64(p, x) lds_add_f64_(p, x, jdet)
__device__ __forceinline__ void lds_add_f64_(double *p, double x, int jdet) {
  if (jdet) (void)__hip_atomic_fetch_add((unsigned long long *)p, (unsigned long long)__builtin_fmin(__builtin_fmax(x * 17592186044416.0, 0.0), 4611686018427387904.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

extern "C" __global__ __launch_bounds__(256, JMINWAVES) void JKERNEL(SmallArgs A) {
  extern __shared__ double lds[];
  // the wavefront index is uniform inside a wavefront, which the compiler cannot see: taken through readfirstlane,
  // everything derived from it (the tile, its pair descriptor, every base address) lives in scalar registers
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jdet = A.nRep >> 16; (void)jdet;
  double *wL = lds;                                          // output-token and match weight tables
#if JMODE == 3
  double *accT = lds + JLDSW;                                // posterior counts of this workgroup's tiles
  double *myL = accT + ((JNTRANS + 1) & ~1) + wv * JWAVEDBL;
#else
  double *myL = lds + JLDSW + wv * JWAVEDBL;
#endif
  int *tokL = (int *)myL;                                    // output tokens entering lane 0, one block of 64 steps
  int *envSL = (int *)(myL + 32), *envEL = envSL + 64;       // envelope rows (inStart, inEnd) entering lane 0, same block
  double *haloL = myL + 96;                                  // halo rows entering lane 0, same block
  (void)envSL; (void)envEL;
#if JMODE == 3
  float *rowL = (float *)(haloL + 64 * JHP) + lane * JROWF;  // this lane's usage sums of the match transitions, by output token
  float *outAcc = (float *)(haloL + 64 * JHP) + 64 * JROWF + (JROWF & 1);   // this wavefront's usage sums of the output-only transitions
#endif
  for (int j = tid; j < JLDSW; j += (int)blockDim.x) wL[j] = (j < JLDSWN) ? A.w[JOFFOUT + j] : 0.0;      // (256 lanes = four tiles, or 64 = one persistent strip)
#if JMODE == 3
  for (int j = tid; j < JNTRANS; j += 256) accT[j] = 0.0;
#endif
  __syncthreads();
  const int tile = A.tileBase + blockIdx.x * (int)(blockDim.x >> 6) + wv;
  if (tile < A.tileEnd) {
    const int4 tl = A.tiles[tile];
    const int pairIdx = tl.x, a = tl.y, b = tl.z;
    const PairDesc pd = A.pairs[pairIdx];
    const SmAux ax = A.aux[pairIdx];
    const int inLen = pd.inLen, outLen = pd.outLen;
    const int NA = (inLen + 64) >> 6;
    const int Te = (outLen + 65) & ~1;
#if JMODE != 3      // (the count sweep always runs launch by launch: see small_sweep -- and has no registers to spare)
    const bool jper = A.flags != nullptr && A.TS >= Te;      // persistent strips: the hand-over is inside the step loop
    if (A.flags && !jper) {
      const int4 dp = A.deps[tile];
      if (lane < 3) { const int d = lane == 0 ? dp.x : (lane == 1 ? dp.y : dp.z); if (d >= 0) sm_wait(A.flags + d, A.err, A.timeoutTicks); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what those tiles stored (halo rows, boundary records) is read below
    }
#else
    const bool jper = false;
#endif
    (void)jper;
    const int t0 = b * A.TS, t1 = min(t0 + A.TS, Te);
    // Backward programs (reversed frame) keep their padding columns in FRONT of the sequence: frame column i of lane c of
    // strip a is then the mirror image of the Forward sweep's lane 63 - c of strip NA - 1 - a, so the count sweep reads
    // every Backward step block whole, by one wavefront, in one run (reversed lane order) instead of two partial runs
    // from two blocks that each share their border cache line with another wavefront.
    const int pad = JREV ? NA * 64 - 1 - inLen : 0;
    const int i = a * 64 + lane - pad;
    const bool colValid = i >= 0 && i <= inLen;
    // lanes of this strip that hold a column of the lattice, and this lane's group of lanes that share a 128-byte line of a
    // store instruction (see the stores below)
    const int cLo = JREV ? max(pad - a * 64, 0) : 0, cHi = JREV ? 63 : min(inLen - a * 64, 63);
    const int gLo = lane & ~(JLINELANES - 1), gHi = lane | (JLINELANES - 1);
    (void)cLo; (void)cHi; (void)gLo; (void)gHi;
    const int *in = A.inTok + pd.inBase, *out = A.outTok + pd.outBase;
    const int it = (colValid && i > 0) ? (JREV ? in[inLen - i] : in[i - 1]) : 0;
    auto tokAt = [&](int o) -> int { return (o >= 1 && o <= outLen) ? (JREV ? out[outLen - o] : out[o - 1]) : 0; };
    cdbl_t wc = (cdbl_t)A.w;
    const int mrow = it * (JNOUT + 1);
    (void)wc; (void)mrow; (void)NA;
/*@WEIGHTS@*/
    const double *haloIn = A.halo + ax.halo + (long long)max(a - 1, 0) * (outLen + 1) * JHP;   // written by strip a-1
    double *haloOut = A.halo + ax.halo + (long long)a * (outLen + 1) * JHP;
    double *bnd = A.bound + ax.bound + ((long long)a * 64 + lane) * JNBD;
    (void)haloIn; (void)haloOut;
#if JMAT
    double *poolPair = A.pool + ax.pool;
#endif
#if JMODE == 2
    unsigned char *tbPair = A.tb + ax.tb;
#endif
#if JMODE == 3
    // the Backward matrix of the pair (stored in the reversed frame of the Backward sweep) and its log-likelihood
    const double *poolB = A.pool + ax.pool;
    const double bLL = A.bwdLL[pairIdx];                          // BackwardMatrix::logLike(), src/backward.cpp:48-50,66
    const double negLL = (bLL > NEG_INF) ? -bLL : NEG_INF;        // -inf likelihood: every term becomes exp(-inf) = 0
    // Cell (i, o) of this lane at step t sits in the Backward sweep's strip NA-1-a, step (outLen - o) + (63 - lane) =
    // outLen + 63 - t, lane 63 - lane: ONE step block for the whole wavefront, read whole (lanes outside the lattice read
    // their mirror slot, which exists and is ignored, instead of a clamped address -- a clamped lane pulls a cache-line
    // sector of its own from each of the block's chunk rows: 47 padding lanes of the last strip cost 1.3x the traffic).
    auto bPtr = [&](int t) -> const double * {
      return poolB + (((long long)(NA - 1 - a) * Te + max(outLen + 63 - t, 0)) * (JNCH * 64) + (63 - lane)) * (JCHB / 8);
    };
#pragma unroll 1
    for (int j = 0; j < JROWF; ++j) rowL[j] = 0.0f;
    for (int j = lane; j < JOUTACC; j += 64) outAcc[j] = 0.0f;
#endif
/*@STATE@*/
    if (b > 0 && !(tl.w & 1)) {      // (bit 0: the block before this one lies outside the envelope and did not run: start from -inf)
/*@LOADBND@*/
    }
#if JENV && JH > 0
    if (b > 0 && (tl.w & 1) && a > 0 && t0 - 1 <= outLen) {
      // The block before this one holds no cell of the envelope and did not run, so there is no boundary record -- but lane 0's
      // DIAGONAL predecessor at step t0, cell (64a - 1, t0 - 1), is not a cell of that block: it is the last column of strip
      // a - 1, whose tile (a - 1, b) ran two launches ago and left it in the halo column (a gapless stretch of an alignment
      // that crosses a strip boundary exactly at a block boundary: path envelope of width 0, quirk Q1's Envelope(seqPair)).
/*@LOADDIAG@*/
    }
#endif
    int ot = tokAt(t0 - 1 - lane);
#if JENV
    // restricted envelope of the pair (src/seqpair.h:75-97): cell (x, y) exists <=> inStart[y] <= x < inEnd[y]; rows travel
    // along the lanes with their output position like the tokens do.  A full-envelope pair of the batch reads [0, inLen+1).
    const int *envS = pd.envBase >= 0 ? A.envStart + pd.envBase : nullptr, *envE = pd.envBase >= 0 ? A.envEnd + pd.envBase : nullptr;
    auto envRow = [&](int o, const int *e, int dflt) -> int { return (e && o >= 0 && o <= outLen) ? e[JREV ? outLen - o : o] : dflt; };
    int es = envRow(t0 - 1 - lane, envS, 0), ee = envRow(t0 - 1 - lane, envE, inLen + 1);
    const int iOrig = JREV ? inLen - i : i;
#endif
#if JH > 0
    if (a == 0) for (int h = 0; h < JH; ++h) haloL[lane * JH + h] = NEG_INF;
#endif
    for (int tb = t0; tb < t1; tb += 64) {
      tokL[lane] = tokAt(tb + lane);
#if JENV
      envSL[lane] = envRow(tb + lane, envS, 0); envEL[lane] = envRow(tb + lane, envE, inLen + 1);
#endif
#if JH > 0
      if (a > 0 && !jper) {
        const int ho = tb + lane;
#pragma unroll
        for (int h = 0; h < JH; ++h) haloL[lane * JH + h] = (ho <= outLen) ? haloIn[(long long)min(ho, outLen) * JH + h] : NEG_INF;
      }
#endif
      wave_sync();
      const int nj = min(64, t1 - tb);
      const int sub = jper ? JSUB : 64;
      for (int j0 = 0; j0 < nj; j0 += sub) {
        const int j1 = min(j0 + sub, nj);
#if JH > 0 && JMODE != 3
        if (jper && a > 0) {
          // halo rows tb + j0 ... tb + j1 - 1: the strip to the left stores row r in its step r + 63
          if (lane < j1 - j0) {
            const int ho = tb + j0 + lane;
#pragma unroll
            for (int h = 0; h < JH; ++h) haloL[(j0 + lane) * JH + h] = (ho <= outLen) ? halo_wait(haloIn + (long long)ho * JH + h, A.err, A.timeoutTicks) : NEG_INF;
          }
          wave_sync();
        }
#endif
        for (int j = j0; j < j1; j += 2) {
/*@STEP0@*/
/*@STEP1@*/
        }
      }
      wave_sync();   // the block buffers are rewritten next
    }
    if (t1 < Te) {
/*@SAVEBND@*/
    }
#if JMODE == 3
/*@FLUSH@*/
#endif
#if JMODE != 3
    if (A.flags) {      // this tile's halo rows and boundary record are written: the tiles that wait for it may go
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) __hip_atomic_store(A.flags + tile, jper ? (unsigned)Te : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
  }
#if JMODE == 3
  __syncthreads();
  {
    double *rep = A.counts + (long long)(blockIdx.x % (A.nRep & 0xffff)) * JNTRANS;
    for (int e = tid; e < JNTRANS; e += 256) {
      if (jdet) {      // 2^-44 in the workgroup -> 2^-36 in global memory, rounded
        const unsigned long long u = ((const unsigned long long *)accT)[e];
        if (u) (void)__hip_atomic_fetch_add((unsigned long long *)rep + e, u >= (1ull << 62) ? (1ull << 62) : (u + 128ull) >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a saturated table entry stays saturated
        continue;
      }
      const double x = accT[e];
      if (x != 0.0) (void)__hip_atomic_fetch_add(rep + e, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}
)MBSM";

static std::string I(long long v) { return std::to_string(v); }

// kernel names tell the sweeps apart in a rocprof trace
const char *small_kernel_name(const SmallProgram &P, int mode, bool materialise) {
  if (mode == SM_COUNT) return "k_small_count";
  if (mode == SM_TB) return "k_small_tb";
  if (mode == SM_MAX) return "k_small_max";
  return P.backward ? "k_small_sum_bwd" : (materialise ? "k_small_sum_mat" : "k_small_sum_roll");
}

// wavefronts per SIMD the kernel is compiled for (register budget 512 / n): the count sweep hides the latency of its
// Backward loads with occupancy; small_jit_get lowers it until the kernel needs no scratch memory
int small_default_minwaves(int mode, bool env) { return std::max(1, env_int_s("MB_SMALL_MINWAVES", mode == SM_COUNT ? (env ? 3 : 4) : 1)); }

std::string small_jit_source(const SmallProgram &P, int mode, bool materialise, bool env, int minWaves) {
  std::ostringstream defs, weights, state, loadb, loadd, saveb, flush;
  const int S = P.S, CB = small_chunk_bytes(S), NCH = small_chunks(S);
  const bool counting = mode == SM_COUNT, tbmode = mode == SM_TB, maxmode = mode == SM_MAX || mode == SM_TB;
  const int rowF = row_floats(P);
  if (minWaves <= 0) minWaves = small_default_minwaves(mode, env);
  defs << "#define JMINWAVES " << minWaves << "\n#define JKERNEL " << small_kernel_name(P, mode, materialise) << "\n#define JS " << S << "\n#define JNIN " << P.nIn << "\n#define JNOUT " << P.nOut << "\n#define JREV " << (P.backward ? 1 : 0)
       << "\n#define JENV " << (env ? 1 : 0) << "\n#define JMODE " << mode << "\n#define JMAT " << (materialise ? 1 : 0) << "\n#define JH " << P.H << "\n#define JHP " << std::max(P.H, 1)
       << "\n#define JNBD " << P.NBD << "\n#define JCHB " << CB << "\n#define JNCH " << NCH << "\n#define JTBSTRIDE " << small_tb_stride(S)
       << "\n#define JNTRANS " << P.nTrans << "\n#define JROWF " << rowF << "\n#define JOUTACC " << outacc_floats(P) << "\n#define JLDSW " << lds_w_doubles(P)
       << "\n#define JLDSWN " << (P.nEntries - P.off[2]) << "\n#define JWAVEDBL " << wave_doubles(P, mode)
       << "\n#define JOFFSIL " << P.off[3] << "\n#define JOFFIN " << P.off[1] << "\n#define JOFFOUT " << P.off[2] << "\n#define JOFFMAT " << P.off[0]
       << "\n#define JSUB " << std::max(2, std::min(64, env_int_s("MB_SMALL_JSUB", 16) & ~1)) << "\n#define JENDSTATE " << P.endState << "\n#define JLINELANES " << (env_int_s("MB_SMALL_STORE_LINES", 1) ? 128 / CB : 64) << "\n";
  for (int k = 0; k < P.nTab[3]; ++k) weights << "    const double wS" << k << " = wc[JOFFSIL + " << k << "];\n";
  for (int k = 0; k < P.nTab[1]; ++k) weights << "    const double wI" << k << " = A.w[JOFFIN + " << (long long)k * (P.nIn + 1) << " + it];\n";
  // persistent state: two sets of cells and of left-neighbour values, alternating by step parity
  for (int p = 0; p < 2; ++p) {
    state << "    double";
    for (int s = 0; s < S; ++s) state << (s ? "," : "") << " c" << p << "_" << s << " = NEG_INF";
    state << ";\n";
    if (!P.needLeft.empty()) {
      state << "    double";
      for (size_t k = 0; k < P.needLeft.size(); ++k) state << (k ? "," : "") << " l" << p << "_" << P.needLeft[k] << " = NEG_INF";
      state << ";\n";
    }
  }
  if (counting) {
    if (P.nTab[3] + P.nTab[1] > 0) {
      state << "    float";
      bool first = true;
      for (int k = 0; k < P.nTab[3]; ++k) { state << (first ? "" : ",") << " aS" << k << " = 0.0f"; first = false; }
      for (int k = 0; k < P.nTab[1]; ++k) { state << (first ? "" : ",") << " aI" << k << " = 0.0f"; first = false; }
      state << ";\n";
    }
    // usage of an output-only transition travels with its output token from lane to lane (one position of the output
    // sequence visits every column of the strip) and is set down when the token leaves lane 63
    for (int k = 0; k < P.nTab[2]; ++k) state << "    float aO" << k << " = 0.0f;\n";
  }
  {  // boundary record: cells of the last step, then the left values of the last step (the next step's diagonal)
    int k = 0;
    for (int s : P.saveCells) { loadb << "      c1_" << s << " = bnd[" << k << "];\n"; saveb << "      bnd[" << k << "] = c1_" << s << ";\n"; ++k; }
    for (int s : P.needDiag) { loadb << "      l1_" << s << " = bnd[" << k << "];\n"; saveb << "      bnd[" << k << "] = l1_" << s << ";\n"; ++k; }
  }
  for (int s : P.needDiag) {   // (needDiag is a subset of needLeft: the halo row holds every state of needLeft, in that order)
    const size_t k = std::find(P.needLeft.begin(), P.needLeft.end(), s) - P.needLeft.begin();
    loadd << "      { const double hv = haloIn[(long long)(t0 - 1) * JH + " << k << "]; l1_" << s << " = lane == 0 ? hv : l1_" << s << "; }\n";
  }
  auto step = [&](int p) {
    const int q = 1 - p;
    std::ostringstream b;
    const std::string cp = "c" + I(p) + "_", cq = "c" + I(q) + "_", lp = "l" + I(p) + "_", lq = "l" + I(q) + "_";
    b << "        {  // step parity " << p << "\n";
    b << "          const int jj = j + " << p << ", t = tb + jj, o = t - lane;\n";
    b << "          const bool active = colValid && o >= 0 && o <= outLen;\n";
    if (materialise || counting)
      b << "          const bool lineOn = gHi >= max(t - outLen, cLo) && gLo <= min(t, cHi);   // some lane of this lane's 128-byte line holds a cell\n";
    const bool lateB = env_int_s("MB_SMALL_BLOAD", 0) != 0;
    auto loadB = [&]() {
      // the Backward supercell of this step's cell: requested at the top of the step, used after its Forward values are done
      b << "          " << (CB == 16 ? "d2" : "double");
      for (int k = 0; k < NCH; ++k) b << (k ? "," : "") << " bq" << k;
      b << ";\n          if (lineOn) { const double *bp = bPtr(t);   // lines without a cell were not written by the Backward sweep\n";
      for (int k = 0; k < NCH; ++k)
        b << "            bq" << k << " = " << (CB == 16 ? "*(const d2 *)(bp + " + I(k * 128) + ")" : "bp[" + I(k * 64) + "]") << ";\n";
      b << "          } else {\n";
      for (int k = 0; k < NCH; ++k) b << "            bq" << k << " = " << (CB == 16 ? "d2{0.0, 0.0}" : "0.0") << ";\n";
      b << "          }\n";
    };
    if (counting && !lateB) loadB();
    if (counting && P.nTab[2] > 0) {
      b << "          if (lane == 63) {\n";
      for (int k = 0; k < P.nTab[2]; ++k) b << "            lds_add_f32(outAcc + " << (long long)k * (P.nOut + 1) << " + ot, aO" << k << ");\n";
      b << "          }\n";
      for (int k = 0; k < P.nTab[2]; ++k) b << "          aO" << k << " = shrf(aO" << k << ", 0.0f);\n";
    }
    b << "          ot = shri(ot, tokL[jj]);\n";
    if (env) b << "          es = shri(es, envSL[jj]); ee = shri(ee, envEL[jj]);\n          const bool inside = active && iOrig >= es && iOrig < ee;\n";
    for (size_t k = 0; k < P.needLeft.size(); ++k)
      b << "          " << lp << P.needLeft[k] << " = shr1(" << cq << P.needLeft[k] << ", haloL[jj * JH + " << k << "]);\n";
    for (int k = 0; k < P.nTab[2]; ++k) b << "          const double wO" << k << " = wL[" << (long long)k * (P.nOut + 1) << " + ot];\n";
    for (int k = 0; k < P.nTab[0]; ++k)
      b << "          const double wM" << k << " = wL[" << (P.off[0] - P.off[2]) + (long long)k * (P.nIn + 1) * (P.nOut + 1) << " + mrow + ot];\n";
    if (counting) b << "          float *rowp = rowL + ot;\n";
    if (tbmode) b << "          unsigned int xw[" << small_tb_stride(S) / 4 << "] = {0};\n";
    for (int d : P.order) {
      const std::vector<SmSlot> &cs = P.cand[d];
      const int n = (int)cs.size();
      b << "          {  // state " << d << "\n";
      for (int k = 0; k < n; ++k) {
        const SmSlot &sl = cs[k];
        const std::string src = (sl.T == 0 ? lq : (sl.T == 1 ? lp : (sl.T == 2 ? cq : cp))) + I(sl.src);
        const std::string w = std::string(sl.T == 0 ? "wM" : (sl.T == 1 ? "wI" : (sl.T == 2 ? "wO" : "wS"))) + I(sl.tab);
        b << "            const double v" << k << " = " << src << " + " << w << ";\n";
      }
      if (n == 0) b << "            double res = NEG_INF;\n";
      else if (maxmode) {
        b << "            double res = v0;\n";
        if (tbmode) b << "            unsigned int x = 0u;\n";
        for (int k = 1; k < n; ++k) {
          if (tbmode) b << "            { const bool g = v" << k << " > res; res = g ? v" << k << " : res; x = g ? " << k << "u : x; }\n";
          else b << "            res = dmax(res, v" << k << ");\n";
        }
        if (tbmode) b << "            xw[" << d / 4 << "] |= x << " << 8 * (d % 4) << ";\n";
      } else if (n == 1) b << "            double res = v0;\n";
      else if (n == 2) b << "            double res = lse2(v0, v1);\n";
      else {
        b << "            double mx = dmax(v0, v1);\n";
        for (int k = 2; k < n; ++k) b << "            mx = dmax(mx, v" << k << ");\n";
        b << "            const double gM = (mx == NEG_INF) ? 0.0 : mx;\n            const float sm = ex2(v0 - gM)";
        for (int k = 1; k < n; ++k) b << " + ex2(v" << k << " - gM)";
        b << ";\n            double res = gM + (double)(__builtin_amdgcn_logf(sm) * SM_LN2);\n";
      }
      if (d == P.seedState) b << "            res = ((i | o) == 0) ? 0.0 : res;   // cell(0,0,start) = 0: no other candidate is finite there\n";
      if (env) b << "            res = inside ? res : NEG_INF;   // cells outside the envelope stay -inf (src/dpmatrix.defs.h:36, dpmatrix.h:142-144)\n";
      b << "            " << cp << d << " = res;\n";
      b << "          }\n";
    }
    // ---- outputs of the step ----
    if (materialise) {
      // stores are switched off in units of WHOLE 128-byte lines: a line with at least one cell of the lattice is written by
      // all of its lanes (the slots of the others exist and are never read), a line with none is not written at all.  With
      // single lanes masked off, the partial cache lines at the two ends of the active range cost more than the bytes saved
      // (scripts/micro/tile_major_probe.hip: 3.3 vs 5.4 TB/s); with everything stored, the skewed start / end of a strip
      // sweep and the padding columns of the last strip are 1.3 x the traffic on 400 x 400 pairs.
      b << "          if (lineOn) {\n            double *dst = poolPair + ((long long)(a * Te + t) * (JNCH * 64) + lane) * (JCHB / 8);\n";
      for (int k = 0; k < NCH; ++k) {
        if (CB == 16) b << "            { d2 v; v.x = " << cp << 2 * k << "; v.y = " << cp << 2 * k + 1 << "; *(d2 *)(dst + " << k * 128 << ") = v; }\n";
        else b << "            dst[" << k * 64 << "] = " << cp << k << ";\n";
      }
      b << "          }\n";
    }
    if (tbmode) {
      b << "          {\n            unsigned int *tp = (unsigned int *)(tbPair + ((long long)(a * Te + t) * 64 + lane) * JTBSTRIDE);\n";
      for (int k = 0; k < small_tb_stride(S) / 4; ++k) b << "            tp[" << k << "] = xw[" << k << "];\n";
      b << "          }\n";
    }
    if (P.H > 0) {
      b << "          if (lane == 63 && a + 1 < NA && o >= 0 && o <= outLen) {\n";
      for (size_t k = 0; k < P.needLeft.size(); ++k) b << "            halo_store(haloOut + (long long)o * JH + " << k << ", " << cp << P.needLeft[k] << ", jper);\n";
      b << "          }\n";
    }
    b << "          if (active && i == inLen && o == outLen) A.loglike[pairIdx] = " << cp << P.endState << ";\n";
    if (counting && lateB) loadB();
    if (counting) {
      // posterior usage of every candidate's transition: exp(F(src) + w + B(dst) - LL), src/backward.cpp:58-87.  The
      // candidate is formed again (one fp64 add) rather than kept alive across the step.
      for (int d : P.order) {
        const std::vector<SmSlot> &cs = P.cand[d];
        if (cs.empty()) continue;
        const std::string B = CB == 16 ? ("bq" + I(d / 2) + (d % 2 ? ".y" : ".x")) : ("bq" + I(d));
        b << "          {  // usage of the transitions into state " << d << "\n";
        b << "            const double bl = " << (env ? "inside" : "active") << " ? (" << B << " + negLL) : NEG_INF;\n";
        for (size_t k = 0; k < cs.size(); ++k) {
          const SmSlot &sl = cs[k];
          const std::string src = (sl.T == 0 ? lq : (sl.T == 1 ? lp : (sl.T == 2 ? cq : cp))) + I(sl.src);
          const std::string w = std::string(sl.T == 0 ? "wM" : (sl.T == 1 ? "wI" : (sl.T == 2 ? "wO" : "wS"))) + I(sl.tab);
          const std::string term = "ex2((" + src + " + " + w + ") + bl)";
          if (sl.T == 3) b << "            aS" << sl.tab << " += " << term << ";\n";
          else if (sl.T == 1) b << "            aI" << sl.tab << " += " << term << ";\n";
          else if (sl.T == 2) b << "            aO" << sl.tab << " += " << term << ";\n";
          else b << "            lds_add_f32(rowp + " << (long long)sl.tab * (P.nOut + 1) << ", " << term << ");\n";
        }
        b << "          }\n";
      }
    }
    b << "        }\n";
    return b.str();
  };
  if (counting) {
    // lane sums -> the workgroup's table.  Silent transitions: the edge is known here; input-token transitions: the edge
    // of this lane's input token; output-token and match rows: one entry per output token.
    for (int k = 0; k < P.nTab[3]; ++k)
      flush << "    { const float r = wave_sum(aS" << k << "); if (lane == 0) lds_add_f64(accT + " << P.eid[(size_t)(P.off[3] + k)] << ", (double)r); }\n";
    for (int k = 0; k < P.nTab[1]; ++k)
      flush << "    { const int e = A.eid[JOFFIN + " << (long long)k * (P.nIn + 1) << " + it]; if (e >= 0) lds_add_f64(accT + e, (double)aI" << k << "); }\n";
    for (int k = 0; k < P.nTab[2]; ++k) flush << "    lds_add_f32(outAcc + " << (long long)k * (P.nOut + 1) << " + ot, aO" << k << ");\n";
    flush << "    wave_sync();\n";
    flush << "#pragma unroll 1\n    for (int j = lane; j < " << P.nTab[2] * (P.nOut + 1) << "; j += 64) {\n"
          << "      const int e = A.eid[JOFFOUT + j]; const float r = outAcc[j];\n"
          << "      if (e >= 0 && r != 0.0f) lds_add_f64(accT + e, (double)r);\n    }\n";
    flush << "#pragma unroll 1\n    for (int k = 0; k < " << P.nTab[0] << "; ++k) {\n#pragma unroll 1\n      for (int x = 1; x <= JNOUT; ++x) {\n"
          << "        const int e = A.eid[JOFFMAT + k * ((JNIN + 1) * (JNOUT + 1)) + mrow + x]; const float r = rowL[k * (JNOUT + 1) + x];\n"
          << "        if (e >= 0 && r != 0.0f) lds_add_f64(accT + e, (double)r);\n      }\n    }\n";
  }
  std::string src = kSmallSkeleton;
  auto replace = [&](const std::string &mark, const std::string &with) {
    const size_t p = src.find(mark);
    if (p != std::string::npos) src.replace(p, mark.size(), with);
  };
  replace("/*@DEFS@*/", defs.str());
  replace("/*@WEIGHTS@*/", weights.str());
  replace("/*@STATE@*/", state.str());
  replace("/*@LOADBND@*/", loadb.str());
  replace("/*@LOADDIAG@*/", loadd.str());
  replace("/*@SAVEBND@*/", saveb.str());
  replace("/*@STEP0@*/", step(0));
  replace("/*@STEP1@*/", step(1));
  replace("/*@FLUSH@*/", flush.str());
  return src;
}

bool small_jit_get(SmallProgram &P, int mode, bool materialise, bool env) {
  SmJit &J = P.jit[mode][materialise ? 1 : 0][env ? 1 : 0];
  if (J.tried) return J.func != nullptr;
  J.tried = true;
  if (!P.ok) return false;
  J.ldsBytes = small_jit_lds_bytes(P, mode);
  if (J.ldsBytes > 160 * 1024) { set_error("small-machine kernel: tables exceed the LDS"); return false; }
  // Registers: every state, neighbour value, weight and usage sum of the machine is a named VGPR; a machine near the
  // family's limit (12-16 states, dozens of tables) does not fit the budget of 3-4 wavefronts per SIMD.  A kernel that
  // spills to SCRATCH memory is not used, in ANY mode: the budget is raised (fewer wavefronts per SIMD) until it needs
  // none, and if even one wavefront per SIMD does not fit, the kernel is reported as not runnable and the caller takes
  // another family (small_can_run).  Why: DESIGN.md section 4.0 "Registers" -- a spilled sweep is slow, and one spilled
  // build gave wrong values.  Metadata that cannot be read counts as "spills" (fail closed).
  std::string code, log, src;
  bool fromCache = false;
  for (int mw = small_default_minwaves(mode, env); mw >= 1; --mw) {
    src = small_jit_source(P, mode, materialise, env, mw);
    if (const char *dump = opt_env("MB_SMALL_JIT_DUMP")) {
      const std::string fn = std::string(dump) + ".m" + I(mode) + (materialise ? ".mat" : ".roll") + (P.backward ? ".bwd" : ".fwd") + ".hip";
      if (FILE *f = fopen(fn.c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    if (!jit_compile(src, "mb_small_jit.hip", code, &log, &fromCache)) {
      if (opt_env("MB_SMALL_JIT_VERBOSE") || opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] hiprtc failed (small family):\n%s\n", log.c_str());
      set_error("run-time compilation of the small-machine kernel failed: " + log.substr(0, 400));
      return false;
    }
    J.scratch = jit_kernel_meta(code, ".private_segment_fixed_size") != 0;
    if (opt_env("MB_SMALL_JIT_VERBOSE")) fprintf(stderr, "[mbhip] small family, mode %d: %d wavefront(s) per SIMD, %lld VGPRs, %lld spilled, scratch %lld bytes\n", mode, mw,
                                                jit_kernel_meta(code, ".vgpr_count"), jit_kernel_meta(code, ".vgpr_spill_count"), jit_kernel_meta(code, ".private_segment_fixed_size"));
    if (!J.scratch || env_int_s("MB_SMALL_ALLOW_SCRATCH", 0)) break;
  }
  if (J.scratch && !env_int_s("MB_SMALL_ALLOW_SCRATCH", 0)) { set_error("small-machine kernel: does not fit the register file (another kernel family takes the machine)"); return false; }
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
    // a cached code object the loader rejects (truncated file, other compiler build): drop it and compile afresh, once
    mod = nullptr;
    if (fromCache) { jit_evict(src); if (!jit_compile(src, "mb_small_jit.hip", code, &log, nullptr) || hipModuleLoadData(&mod, code.data()) != hipSuccess) mod = nullptr; }
    if (!mod) { set_error("small-machine kernel: hipModuleLoadData failed"); return false; }
  }
  if (hipModuleGetFunction(&fn, mod, small_kernel_name(P, mode, materialise)) != hipSuccess) { (void)hipModuleUnload(mod); set_error("small-machine kernel: entry point missing"); return false; }
  (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  J.module = mod; J.func = fn;
  return true;
}

// Can this sweep run on this family?  The kernel is built here if it has not been (it would be built by the sweep anyway).
// The count sweep keeps one usage row per lane in LDS: machines with many output-token / match tables do not fit.
bool small_can_run(SmallProgram &P, int mode, bool materialise, bool env) {
  if (!P.ok) return false;
  if (mode == SM_COUNT && small_jit_lds_bytes(P, SM_COUNT) > 96 * 1024) return false;
  return small_jit_get(P, mode, materialise, env);
}
bool small_count_fits(SmallProgram &P, bool env) { return small_can_run(P, SM_COUNT, false, env); }

// ---- one sweep: the wavefront of tiles --------------------------------------------------------------------------------------
struct SmallArgsHost {   // must match SmallArgs in the generated source
  const PairDesc *pairs; const int *inTok; const int *outTok;
  const int4 *tiles; int tileBase, tileEnd, TS, nRep;
  double *pool; unsigned char *tb; double *halo; double *bound; const SmAux *aux;
  double *loglike; const double *w; const int *eid; const double *bwdLL; double *counts;
  const int *envStart; const int *envEnd;
  const int4 *deps; unsigned *flags; unsigned *err; long long timeoutTicks;
};

// Steps per tile: the longest tile that still leaves ~12 tiles per CU in an average launch; 64 (the minimum: a tile must
// not run ahead of the halo rows the strip to its left has produced) when the batch cannot fill the chip anyway.
static int pick_tile_steps(const std::vector<PairDesc> &pairs) {
  const int forced = env_int_s("MB_SMALL_TS", 0);
  if (forced >= 64 && forced % 64 == 0) return forced;
  int best = 64;
  for (int TS : {128, 256, 512}) {
    long long tiles = 0; int nLaunch = 0;
    for (const PairDesc &pd : pairs) {
      const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
      tiles += (long long)NA * NB;
      nLaunch = std::max(nLaunch, 2 * (NA - 1) + NB);
    }
    if (tiles / std::max(nLaunch, 1) >= 3072) best = TS;
  }
  return best;
}

int small_sweep(SmallProgram &P, int mode, bool materialise, const SmSweep &sw, hipStream_t st) {
  const std::vector<PairDesc> &pairs = *sw.pairs;
  if (pairs.empty()) return 0;
  const bool timing = opt_env("MB_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tPrev = now();
  auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[mbhip]   sweep %-24s %7.2f ms\n", what, t - tPrev); tPrev = t; } };
  if (!small_jit_get(P, mode, materialise, sw.d_envStart != nullptr)) return 1;
  lap("kernel lookup / jit");
  const SmJit &J = P.jit[mode][materialise ? 1 : 0][sw.d_envStart != nullptr ? 1 : 0];
  // PERSISTENT STRIPS: every strip of every pair is one wavefront of ONE grid and sweeps its whole strip; the halo column between two
  // strips is its own hand-over (HALO_EMPTY in the generated source).  For batches that cannot fill the chip -- a single pair, the
  // unchanged `for (seqPair : data.seqPairs)` loops of target/boss.cpp:796-833, src/api.cpp:31-66 -- the sweep is then bound by the
  // lattice's own critical path -- a strip lags its left neighbour by 63 + JSUB steps -- instead of 2 NA + NB launches of 64-step tiles
  // (two blocks of lag and 3 us of tile set-up per block): one 1 kb x 1 kb dnapsw pair 1.19 -> 0.84 ms (Forward), 0.97 -> 0.69 ms
  // (Viterbi) on the same box, bit-identical (scripts/single_pair_probe.py; hand-over blocks of 8 / 16 / 32 steps: 0.94 / 0.90 / 0.95 ms
  // with four strips per workgroup, 0.84 with one: a step of 16 concurrent strips is 0.36 us, so 1 088 + 15 x 79 steps are 0.82 ms).  Chosen when all strips are co-resident with one workgroup (four strips) per CU at most
  // (<= 1 024 strips; a larger batch fills the chip with tiles and is bound by vector issue either way); not for the count sweep (it
  // adds into the caller's accumulators as it goes: a sweep that gave up could not simply be run again) nor with dead tiles.
  // MB_SMALL_ONE_LAUNCH: 0 never, 2 always, 1 the 64-step TILES of a sweep in one grid with done-flags (measured: no faster than the
  // launches), default: the rule above.  A wait that runs out (MB_SMALL_ONE_LAUNCH_TIMEOUT_S; a shared device) latches both forms off.
  static bool oneLaunchOff = false;
  const int oneWant = env_int_s("MB_SMALL_ONE_LAUNCH", -1);
  long long nStrips = 0;
  int chain = 0;      // launches of the tile form (64-step tiles): a pair of one or two tiles has no chain to shorten (config 1's 50 x 50 pair: 85 us per call, 95 through the strips)
  for (const PairDesc &pd : pairs) { nStrips += small_strips(pd.inLen); chain = std::max(chain, 2 * (small_strips(pd.inLen) - 1) + (small_steps(pd.outLen) + 63) / 64); }
  const bool persist = (oneWant == 2 || (oneWant < 0 && nStrips <= 1024 && chain >= 4)) && !oneLaunchOff && mode != SM_COUNT && !(sw.d_envStart != nullptr && sw.h_envStart != nullptr);
  const int TS = persist ? (1 << 20) : pick_tile_steps(pairs);
  // The tile lists depend on the pairs' shapes (and envelopes) only: built and uploaded once per batch chunk and sweep
  // direction, reused by every later sweep.  With restricted envelopes a tile none of whose cells lies inside its pair's
  // envelope is not launched at all: its cells stay -inf (the halo columns are pre-filled with -inf, the next block of
  // the strip is told to start from -inf instead of loading the boundary record).
  SmTileCache local;
  SmTileCache *cached = P.backward ? sw.tileCacheBwd : sw.tileCacheFwd;
  SmTileCache &tc = cached ? *cached : local;
  const bool env = sw.d_envStart != nullptr, skipDead = env && sw.h_envStart != nullptr;
  if (!tc.d_tiles || tc.TS != TS) {
    if (tc.d_tiles) { (void)hipFree(tc.d_tiles); tc.d_tiles = nullptr; }
    int nLaunch = 0;
    for (const PairDesc &pd : pairs) {
      const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
      nLaunch = std::max(nLaunch, 2 * (NA - 1) + NB);
    }
    // live[p][a * NB + b]
    std::vector<std::vector<char>> live(pairs.size());
    for (size_t p = 0; p < pairs.size(); ++p) {
      const PairDesc &pd = pairs[p];
      const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
      live[p].assign((size_t)NA * NB, (skipDead && pd.envBase >= 0) ? 0 : 1);
      if (!(skipDead && pd.envBase >= 0)) continue;
      const int pad = P.backward ? NA * 64 - 1 - pd.inLen : 0;
      for (int o = 0; o <= pd.outLen; ++o) {
        const int es = sw.h_envStart[pd.envBase + o], ee = sw.h_envEnd[pd.envBase + o];   // original coordinates: columns [es, ee) of row o
        if (ee <= es) continue;
        const int fo = P.backward ? pd.outLen - o : o;
        // frame columns of the row, as global lane indices (strip * 64 + lane)
        const int g0 = (P.backward ? pd.inLen - (ee - 1) : es) + pad, g1 = (P.backward ? pd.inLen - es : ee - 1) + pad;
        for (int a = g0 >> 6; a <= (g1 >> 6); ++a) {
          const int c0 = std::max(g0, a * 64) - a * 64, c1 = std::min(g1, a * 64 + 63) - a * 64;
          for (int b = (fo + c0) / TS; b <= (fo + c1) / TS; ++b) live[p][(size_t)a * NB + b] = 1;
        }
      }
    }
    std::vector<long long> cnt(nLaunch + 1, 0);
    for (size_t p = 0; p < pairs.size(); ++p) {
      const PairDesc &pd = pairs[p];
      const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
      for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) if (live[p][(size_t)a * NB + b]) cnt[2 * a + b]++;
    }
    tc.off.assign(nLaunch + 1, 0);
    for (int l = 0; l < nLaunch; ++l) tc.off[l + 1] = tc.off[l] + cnt[l];
    if (tc.off[nLaunch] > 0x7fffffffLL) { set_error("too many tiles in one sweep"); return 1; }
    std::vector<int4> tiles((size_t)tc.off[nLaunch]), deps((size_t)tc.off[nLaunch], make_int4(-1, -1, -1, -1));
    {
      std::vector<long long> fill(tc.off.begin(), tc.off.end() - 1);
      std::vector<std::vector<int>> pos(pairs.size());      // position of every live tile in the list
      for (size_t p = 0; p < pairs.size(); ++p) {
        const PairDesc &pd = pairs[p];
        const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
        pos[p].assign((size_t)NA * NB, -1);
        for (int a = 0; a < NA; ++a)
          for (int b = 0; b < NB; ++b)
            if (live[p][(size_t)a * NB + b]) {
              pos[p][(size_t)a * NB + b] = (int)fill[2 * a + b];
              tiles[(size_t)fill[2 * a + b]++] = make_int4((int)p, a, b, (b > 0 && !live[p][(size_t)a * NB + b - 1]) ? 1 : 0);
            }
      }
      // what a tile reads from other tiles (the one-launch form): the boundary record of the block before it, the halo rows of the
      // strip to its left written by that strip's blocks b and b + 1 (a tile's rows reach 63 steps past its own)
      for (size_t p = 0; p < pairs.size(); ++p) {
        const PairDesc &pd = pairs[p];
        const int NA = small_strips(pd.inLen), NB = (small_steps(pd.outLen) + TS - 1) / TS;
        for (int a = 0; a < NA; ++a)
          for (int b = 0; b < NB; ++b) {
            const int at = pos[p][(size_t)a * NB + b];
            if (at < 0) continue;
            int4 d = make_int4(-1, -1, -1, -1);
            if (b > 0) d.x = pos[p][(size_t)a * NB + b - 1];
            if (a > 0) { d.y = pos[p][(size_t)(a - 1) * NB + b]; if (b + 1 < NB) d.z = pos[p][(size_t)(a - 1) * NB + b + 1]; }
            deps[(size_t)at] = d;
          }
      }
    }
    lap("tile lists");
    if (!hip_ok(hipMalloc(&tc.d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(int4)), "hipMalloc(tile list)")) return 1;
    if (h2d_large(tc.d_tiles, tiles.data(), tiles.size() * sizeof(int4)) || !hip_ok(hipStreamSynchronize(st), "H2D tile list")) {
      (void)hipFree(tc.d_tiles); tc.d_tiles = nullptr;
      return 1;
    }
    if (tc.d_deps) { (void)hipFree(tc.d_deps); tc.d_deps = nullptr; }
    if (tc.d_flags) { (void)hipFree(tc.d_flags); tc.d_flags = nullptr; }
    if (hipMalloc(&tc.d_deps, std::max<size_t>(deps.size(), 1) * sizeof(int4)) != hipSuccess || hipMalloc(&tc.d_flags, (std::max<size_t>(deps.size(), 1) + 1) * sizeof(unsigned)) != hipSuccess ||
        h2d_large(tc.d_deps, deps.data(), deps.size() * sizeof(int4)) || !hip_ok(hipStreamSynchronize(st), "H2D tile dependencies")) {
      (void)hipGetLastError();      // (without them the sweep runs launch by launch)
      if (tc.d_deps) { (void)hipFree(tc.d_deps); tc.d_deps = nullptr; }
      if (tc.d_flags) { (void)hipFree(tc.d_flags); tc.d_flags = nullptr; }
    }
    tc.TS = TS;
  }
  if (skipDead && sw.haloDoubles > 0 && launch_fill_neg_inf(sw.d_halo, sw.haloDoubles, st)) return 1;   // halo rows of tiles that do not run
  const int nLaunch = (int)tc.off.size() - 1;
  const std::vector<long long> &off = tc.off;
  const int4 *d_tiles = (const int4 *)tc.d_tiles;
  lap("tile list upload");
  SmallArgsHost A{};
  A.pairs = sw.d_pairs; A.inTok = sw.d_in; A.outTok = sw.d_out; A.tiles = d_tiles; A.TS = TS; A.nRep = std::max(sw.nRep, 1) | (g_deterministic ? 1 << 16 : 0);
  A.pool = sw.d_pool; A.tb = sw.d_tb; A.halo = sw.d_halo; A.bound = sw.d_bound; A.aux = sw.d_aux; A.loglike = sw.d_loglike;
  A.w = P.d_w; A.eid = P.d_eid; A.bwdLL = sw.d_bwdLL; A.counts = sw.d_counts;
  A.envStart = sw.d_envStart; A.envEnd = sw.d_envEnd;
  bool ok = true;
  // (MB_SMALL_ONE_LAUNCH=1, the tiles of the sweep in one grid: 47 launches 1.011 ms, one launch 1.033 ms on a 1 kb x 1 kb dnapsw pair --
  //  the hand-over by whole 64-step blocks makes a strip lag the one to its left by two blocks, with one launch or with 47)
  const long long nTiles = off[nLaunch];
  // (not the count sweep: it adds into the caller's accumulators as it goes, so a sweep that gave up could not simply be run again)
  const bool oneLaunch = tc.d_deps && tc.d_flags && nTiles > 0 && !oneLaunchOff && oneWant != 0 && mode != SM_COUNT &&
                         (oneWant == 1 || persist) && nTiles <= (1 << 20);
  if (oneLaunch) {
    if (persist && sw.haloDoubles > 0 && !hip_ok(hipMemsetAsync(sw.d_halo, 0xFF, (size_t)sw.haloDoubles * sizeof(double), st), "memset(halo sentinel)")) return 1;
    unsigned *flags = (unsigned *)tc.d_flags;      // [nTiles] done words + the error word behind them
    if (!hip_ok(hipMemsetAsync(flags, 0, ((size_t)nTiles + 1) * sizeof(unsigned), st), "memset(tile flags)")) return 1;
    A.deps = (const int4 *)tc.d_deps; A.flags = flags; A.err = flags + nTiles;
    A.timeoutTicks = (long long)std::max(1, env_int_s("MB_SMALL_ONE_LAUNCH_TIMEOUT_S", 20)) * 100000000ll;
    if (const int us = env_int_s("MB_SMALL_ONE_LAUNCH_TIMEOUT_US", 0)) A.timeoutTicks = (long long)std::max(us, 1) * 100ll;      // (the tests' way to a time-out: a microsecond)
    A.tileBase = 0; A.tileEnd = (int)nTiles;
    void *args[] = {&A};
    ++g_last_launches;
    // (persistent strips: ONE strip per workgroup -- a wavefront alone on its CU issues a Forward step in 0.29 us, four strips sharing a CU in 0.36)
    const unsigned wpb = persist && env_int_s("MB_SMALL_STRIP_PER_WG", 1) ? 1u : 4u;
    ok = hipModuleLaunchKernel((hipFunction_t)J.func, (unsigned)((nTiles + wpb - 1) / wpb), 1, 1, 64 * wpb, 1, 1, (unsigned)J.ldsBytes, st, args, nullptr) == hipSuccess;
    unsigned e = 0;
    ok = ok && hip_ok(hipGetLastError(), "small tile launch") && hip_ok(hipMemcpyAsync(&e, flags + nTiles, sizeof(e), hipMemcpyDeviceToHost, st), "tile status") && hip_ok(hipStreamSynchronize(st), "small tile kernel");
    lap("one launch");
    if (ok && !e) { if (!cached && local.d_tiles) { (void)hipFree(local.d_tiles); if (local.d_deps) (void)hipFree(local.d_deps); if (local.d_flags) (void)hipFree(local.d_flags); } return 0; }
    if (!ok) { set_error("small-machine kernel launch failed"); return 1; }
    oneLaunchOff = true;
    fprintf(stderr, "[mbhip] WARNING: a tile of a one-launch sweep waited longer than MB_SMALL_ONE_LAUNCH_TIMEOUT_S for the tiles it reads from (is the device shared?): the sweep is run again launch by launch, and so are the sweeps that follow\n");
    A.deps = nullptr; A.flags = nullptr; A.err = nullptr;
    if (persist) { if (!cached && local.d_tiles) { (void)hipFree(local.d_tiles); if (local.d_deps) (void)hipFree(local.d_deps); if (local.d_flags) (void)hipFree(local.d_flags); } return small_sweep(P, mode, materialise, sw, st); }      // (with the tiles of the launch-by-launch form)
    if (skipDead && sw.haloDoubles > 0 && launch_fill_neg_inf(sw.d_halo, sw.haloDoubles, st)) return 1;
  }
  for (int l = 0; l < nLaunch && ok; ++l) {
    const long long nt = off[l + 1] - off[l];
    if (nt <= 0) continue;
    ++g_last_launches;
    A.tileBase = (int)off[l]; A.tileEnd = (int)off[l + 1];
    void *args[] = {&A};
    const unsigned grid = (unsigned)((nt + 3) / 4);
    ok = hipModuleLaunchKernel((hipFunction_t)J.func, grid, 1, 1, 256, 1, 1, (unsigned)J.ldsBytes, st, args, nullptr) == hipSuccess;
  }
  lap("launches");
  if (!ok) set_error("small-machine kernel launch failed");
  ok = ok && hip_ok(hipGetLastError(), "small tile launch") && hip_ok(hipStreamSynchronize(st), "small tile kernels");
  lap("stream synchronize");
  if (!cached && local.d_tiles) { (void)hipFree(local.d_tiles); if (local.d_deps) (void)hipFree(local.d_deps); if (local.d_flags) (void)hipFree(local.d_flags); }
  return ok ? 0 : 1;
}

}  // namespace mb

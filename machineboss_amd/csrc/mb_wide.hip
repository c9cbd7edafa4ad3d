// mb_wide.hip -- kernel family for LARGE ONE-TAPE machines (profile-HMM generators composed with transducers:
// BASELINE config 5, `boss --hmmer x.hmm preset... --output-fasta`).
//
// A one-tape machine has no input alphabet, so the lattice of a sequence is (outLen+1) columns of nStates cells and a
// column depends on the previous column only (MappedForwardMatrix::fill with inLen = 0, src/forward.defs.h:23-49):
// there is no anti-diagonal to spread over workgroups, the parallelism inside one sequence is the STATES of one column
// (thousands), and the cost is the dependency depth of a column -- the silent levels (246 for a 20-node profile
// composed with simple_introns . translate . dnapsw, 1000+ for the whole fn3 profile).
//
//   * one workgroup (1024 lanes) per sequence; the previous and the current column live in LDS (fp64), or in an
//     L2-resident scratch vector when 2 x nStates doubles do not fit the 160 KB of a CU;
//   * a state is finalised by a GROUP of 1..64 lanes: each lane folds every g-th candidate of the state, the group is
//     reduced with wavefront shuffles (max, or max + sum of exp in the log-sum-exp semiring), the first lane stores;
//     group sizes are chosen per stage so that the candidates of the stage spread over all lanes of the workgroup;
//   * Forward / Backward: the silent levels are grouped into stages, each closed transitively on the host (the same
//     construction as the tiled family, mb_medium.hip) -- one __syncthreads() per stage instead of one per level.  Stage
//     boundaries are adaptive (levels join a stage while its closure stays within a budget of candidates: the thin runs
//     of a profile's delete chain end up in few stages), the shape is picked by a cost model fitted on the box.
//     Viterbi: level by level, one rounded add per edge, bit-identical to the reference; stages that fit the first
//     wavefront follow each other without a barrier.
//   * candidate records are streamed from L2 ([slot][lane] order: one coalesced load per lane and slot, 8 slots ahead).
//   * log-sum-exp sweeps of machines whose fp64 columns exceed the LDS run in fp32 relative to a per-column fp64
//     reference (k_wide_sum32), current column in LDS, previous one in L2 if need be.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>
#include <type_traits>

#include "mb_wide.h"
#include "mb_wide_jit.h"
#include "mb_device_math.h"

namespace mb {

static constexpr double W_NEG_BIG = -1e300;       // finite stand-in for -inf in the running maximum (avoids inf - inf)
static constexpr uint32_t W_IDX_MASK = 0x03ffffffu;
static constexpr uint32_t W_NO_DST = 0x03ffffffu;
static constexpr long long WIDE_STAGES_DP = 1ll << 30;      // wide_nodes(K = -WIDE_STAGES_DP): stage boundaries by dynamic programming
static const size_t WIDE_LDS_MAX = 160 * 1024;

// ------------------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------------------
// v_max_f64 as is: the operands here are sums of finite weights or -inf, never NaN, so the quieting moves the compiler puts
// in front of fmax() (one v_max_f64 x, x per operand) buy nothing
__device__ __forceinline__ double wide_max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// one candidate folded into a lane's running (max, sum of exp relative to max): exactly one of the two exponentials of
// the usual online update is exp(0), so a single v_exp_f32 of -|v - m| serves both cases
// exp(x) of an fp64 difference x <= 0 through v_exp_f32: the multiplication by log2(e) is done in fp64, BEFORE the conversion.  __expf
// multiplies in fp32 by a log2(e) that is 1.3e-8 too small, i.e. returns exp(x (1 - 1.3e-8)): a relative BIAS of 1.3e-8 |x| on every
// non-dominant term of a log-sum-exp.  A bias does not average out -- it accumulates at a rate that differs between states (a
// chain of self-loop dominated states gathers less of it than the match states beside it): at 50 000 columns the posterior counts
// of whole groups of transitions sat 7.6e-5 from the exact oracle (round 5, scripts/count_accuracy_onetape.py).
__device__ __forceinline__ float wide_exp_diff(double x) { return __builtin_amdgcn_exp2f((float)(x * 1.4426950408889634)); }
template <int MODE>
__device__ __forceinline__ void wide_fold(double &m, float &s, double v, float sv) {
  if (MODE == MB_VITERBI) { m = dmax(m, v); return; }
  const float e = wide_exp_diff(-fabs(v - m));
  const bool up = v > m;
  s = __fmaf_rn(up ? s : sv, e, up ? sv : s);
  m = wide_max_raw(m, v);      // (fmax() costs two more v_max_f64 per fold: half of the maxima of the Forward sweep's ISA were x = max(x, x))
}

// value of lane (l ^ H) of the lane group: H = 1, 2 quad permutes, H = 4, 8 half-row / row mirrors (equivalent to the
// xor once the smaller steps have made quads / half-rows uniform), H = 16, 32 through the LDS crossbar
template <int H>
__device__ __forceinline__ int wide_xor_lane(int v) {
  if (H == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);       // quad_perm:[1,0,3,2]
  if (H == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);       // quad_perm:[2,3,0,1]
  if (H == 4) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);      // row_half_mirror
  if (H == 8) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);      // row_mirror
  return __shfl_xor(v, H, 64);
}

// lane-group reduction, all lanes of the group end with the group's result: butterfly over the maxima, one rescale of
// the lane's own sum to the group maximum, butterfly over the sums
template <int MODE, int H>
__device__ __forceinline__ void wide_max_step(double &m, int g) {
  const double mo = __hiloint2double(wide_xor_lane<H>(__double2hiint(m)), wide_xor_lane<H>(__double2loint(m)));
  if (MODE == MB_VITERBI) { const double mx = wide_max_raw(m, mo); m = (H < g) ? mx : m; }      // (one v_max_f64 + selects; a compare-and-select maximum is two more)
  else if (H < g) m = wide_max_raw(m, mo);
}
template <int H>
__device__ __forceinline__ void wide_sum_step(float &s, int g) {
  const float so = __int_as_float(wide_xor_lane<H>(__float_as_int(s)));
  if (H < g) s += so;
}
template <int MODE>
__device__ __forceinline__ void wide_group_reduce(double &m, float &s, int g, int gWave) {
  const double own = m;
  if (gWave > 1) wide_max_step<MODE, 1>(m, g);
  if (gWave > 2) wide_max_step<MODE, 2>(m, g);
  if (gWave > 4) wide_max_step<MODE, 4>(m, g);
  if (gWave > 8) wide_max_step<MODE, 8>(m, g);
  if (gWave > 16) wide_max_step<MODE, 16>(m, g);
  if (gWave > 32) wide_max_step<MODE, 32>(m, g);
  if (MODE == MB_VITERBI) return;
  s *= wide_exp_diff(own - m);                      // own <= m; an empty lane (own = W_NEG_BIG, s = 0) stays 0
  if (gWave > 1) wide_sum_step<1>(s, g);
  if (gWave > 2) wide_sum_step<2>(s, g);
  if (gWave > 4) wide_sum_step<4>(s, g);
  if (gWave > 8) wide_sum_step<8>(s, g);
  if (gWave > 16) wide_sum_step<16>(s, g);
  if (gWave > 32) wide_sum_step<32>(s, g);
}

// FAST: all vector entries of a workgroup are addressable with 16 bits, and a record carries the source index for both
// parities of the column (the two state vectors swap roles every column): src = index(even column) | index(odd) << 16.
template <int MODE, bool GV, bool FAST>
__global__ __launch_bounds__(1024) void k_wide_sweep(WideDev P1, const PairDesc *__restrict__ pairs1, const int *__restrict__ outTok,
                                                     double *__restrict__ pool1, double *__restrict__ loglike1, double *__restrict__ scratch1,
                                                     WideDev P2, WideSecond X) {
  extern __shared__ double wlds[];
  // a fused launch runs a second sweep (another program, other pairs) in the workgroups from X.nFirst on
  const bool second = blockIdx.x >= X.nFirst;
  const WideDev P = second ? P2 : P1;
  const PairDesc *pairs = second ? X.pairs : pairs1;
  double *pool = second ? X.pool : pool1, *loglike = second ? nullptr : loglike1, *scratch = second ? (double *)X.scratch : scratch1;
  const unsigned bid = blockIdx.x - (second ? X.nFirst : 0u);
  const PairDesc pd = pairs[bid];
  const int tid = threadIdx.x, W = P.W, S = P.S, NV = P.NV;
  const int outLen = P.inputTape ? pd.inLen : pd.outLen, nA = P.nA, n = P.nA + P.nB;   // the one tape the machine has
  double *V = GV ? scratch + (size_t)bid * (size_t)(2 * NV + P.NX) : wlds;
  for (int k = tid; k < 2 * NV + P.NX; k += W) V[k] = -INFINITY;
  __syncthreads();
  if (tid == 0) V[S + 1] = 0.0;                     // the seed, read by the first column only
  __syncthreads();
  int prevOff = 0, curOff = NV;
  const int extraOff = 2 * NV;
  const int *out = outTok + (P.inputTape ? pd.inBase : pd.outBase);
  double *cells = pool ? pool + pd.cellBase : nullptr;
  auto tokOf = [&](int c) -> int {                  // output token the column c of the sweep consumes
    if (c > outLen) return 0;
    const int o = P.backward ? outLen - c : c;
    return P.backward ? (o < outLen ? out[o] : 0) : (o ? out[o - 1] : 0);
  };
  // the record stream: a wave-uniform cursor (segment A of this column's token, then segment B, then segment A of the
  // next column's token, ...) runs WIDE_RING slots ahead of the slot being folded; slot j sits in q[j % WIDE_RING]
  const WideRec *cursor = P.segA + (size_t)tokOf(0) * P.strideA;
  int jn = 0, cc = 0, tokAhead = tokOf(1);          // slot and column of the cursor, token of the column after it
  WideRec q[WIDE_RING];
  auto fetch = [&](WideRec &dst) {
    dst = cursor[tid];
    cursor += W; ++jn;
    if (jn == nA) cursor = P.segB;
    if (jn == n) { jn = 0; ++cc; cursor = P.segA + (size_t)tokAhead * P.strideA; tokAhead = tokOf(cc + 1); }
  };
#pragma unroll
  for (int k = 0; k < WIDE_RING; ++k) fetch(q[k]);
  for (int c = 0; c <= outLen; ++c) {
    const int o = P.backward ? outLen - c : c;
    const int shift = (c & 1) * 16;
    auto at = [&](const WideRec &rc) -> int {
      if (FAST) return (int)((rc.src >> shift) & 0xffffu);
      const uint32_t sel = rc.src >> 30, idx = rc.src & 0x3fffffffu;
      return (sel == 0 ? curOff : (sel == 1 ? extraOff : prevOff)) + (int)idx;
    };
    double m = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG;
    float s = 0.0f;
    for (int j0 = 0; j0 < n; j0 += WIDE_RING) {
#pragma unroll
      for (int k = 0; k < WIDE_RING; ++k) {
        const WideRec rc = q[k];
        fetch(q[k]);
        wide_fold<MODE>(m, s, V[at(rc)] + rc.w, 1.0f);
        const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)rc.pad);
        if (flags & 0x80000000u) {
          const uint32_t dst = rc.pad;
          const int g = 1 << ((dst >> 26) & 7);
          // groups are laid out by decreasing size: the first lane of a wavefront carries the wavefront's largest group
          const int gWave = 1 << ((flags >> 26) & 7);
          if (gWave > 1) wide_group_reduce<MODE>(m, s, g, gWave);
          if ((dst & W_IDX_MASK) != W_NO_DST) {
            const double res = (MODE == MB_VITERBI) ? m : (s > 0.0f ? m + (double)__log2f(s) * 0.6931471805599453 : -INFINITY);      // (ln 2 in fp64: as a float it is 2.7e-9 too large, a bias that a column of hundreds of levels adds up)
            V[((dst >> 29) & 1 ? extraOff : curOff) + (int)(dst & W_IDX_MASK)] = res;
          }
          m = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG;
          s = 0.0f;
          if (flags & 0x40000000u) __syncthreads();
        }
      }
    }
    // the last round of a column always synchronises: the column is complete here
    if (cells && (!P.lastOnly || c == outLen)) {
      double *col = P.lastOnly ? cells : cells + (long long)o * S;
      for (int k = tid; k < S; k += W) col[k] = V[curOff + k];
    }
    if (tid == 0) V[prevOff + S + 1] = -INFINITY;   // the seed is spent (this vector is the next column's `cur`)
    const int t = prevOff; prevOff = curOff; curOff = t;
  }
  if (loglike && tid == 0) loglike[bid] = V[prevOff + P.resultIdx];
}

// ---- levelled max programs walked phase by phase (see WideVitDev) ----------------------------------------------------------
// one slot of a round: fold the lane's candidate; on the round's last slot reduce the lane groups and store
__device__ __forceinline__ void wide_vit_slot(const WideRec rc, double *V, int shift, int curOff, int extraOff, double &m) {
  m = __builtin_fmax(m, V[(rc.src >> shift) & 0xffffu] + rc.w);      // v_max_f64: operands are never NaN, the maximum is exact
  const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)rc.pad);
  if (flags & 0x80000000u) {
    const uint32_t dst = rc.pad;
    const int g = 1 << ((dst >> 26) & 7), gWave = 1 << ((flags >> 26) & 7);
    if (gWave > 1) { float s = 0.0f; wide_group_reduce<MB_VITERBI>(m, s, g, gWave); }
    if ((dst & W_IDX_MASK) != W_NO_DST) V[((dst >> 29) & 1 ? extraOff : curOff) + (int)(dst & W_IDX_MASK)] = m;
    m = -INFINITY;
    if (flags & 0x40000000u) __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void k_wide_viterbi(WideDev P, WideVitDev Q, const PairDesc *__restrict__ pairs, const int *__restrict__ outTok,
                                                       double *__restrict__ pool, double *__restrict__ loglike) {
  extern __shared__ double wlds[];
  const unsigned bid = blockIdx.x;
  const PairDesc pd = pairs[bid];
  const int tid = threadIdx.x, W = P.W, S = P.S, NV = P.NV;
  const int outLen = P.inputTape ? pd.inLen : pd.outLen;
  double *V = wlds;
  for (int k = tid; k < 2 * NV + P.NX; k += W) V[k] = -INFINITY;
  __syncthreads();
  if (tid == 0) V[S + 1] = 0.0;                     // the seed, read by the first column only
  __syncthreads();
  int prevOff = 0, curOff = NV;
  const int extraOff = 2 * NV;
  const int *out = outTok + (P.inputTape ? pd.inBase : pd.outBase);
  double *cells = pool ? pool + pd.cellBase : nullptr;
  const bool first = (tid >> 6) == 0;               // wave-uniform
  for (int c = 0; c <= outLen; ++c) {
    const int o = P.backward ? outLen - c : c;
    const int tok = c > outLen ? 0 : (P.backward ? (o < outLen ? out[o] : 0) : (o ? out[o - 1] : 0));
    const int shift = (c & 1) * 16;
    double m = -INFINITY;
    for (int ph = 0; ph < Q.nPhases; ++ph) {
      const WidePhase h = Q.phase[ph];
      if (!h.thin) {
        // every wavefront: W-lane slots, eight in flight (padded to whole rings on the host; the slack behind the stream is readable)
        const WideRec *p = (h.inA ? Q.wideA + (size_t)tok * Q.strideWideA : Q.wideB) + (size_t)h.off * W + tid;
        WideRec q[WIDE_RING];
#pragma unroll
        for (int k = 0; k < WIDE_RING; ++k) q[k] = p[(size_t)k * W];
        for (int j0 = 0; j0 < h.nSlots; j0 += WIDE_RING) {
#pragma unroll
          for (int k = 0; k < WIDE_RING; ++k) {
            const WideRec rc = q[k];
            q[k] = p[(size_t)(j0 + WIDE_RING + k) * W];
            wide_vit_slot(rc, V, shift, curOff, extraOff, m);
          }
        }
      } else {
        if (first) {
          // the first wavefront alone: 64-lane slots, sixteen in flight; its LDS traffic is ordered, so a level's stores are
          // seen by the next level's loads without a barrier
          const WideRec *p = (h.inA ? Q.thinA + (size_t)tok * Q.strideThinA : Q.thinB) + (size_t)h.off * 64 + tid;
          WideRec q[WIDE_THIN_RING];
#pragma unroll
          for (int k = 0; k < WIDE_THIN_RING; ++k) q[k] = p[(size_t)k * 64];
          for (int j0 = 0; j0 < h.nSlots; j0 += WIDE_THIN_RING) {
#pragma unroll
            for (int k = 0; k < WIDE_THIN_RING; ++k) {
              const WideRec rc = q[k];
              q[k] = p[(size_t)(j0 + WIDE_THIN_RING + k) * 64];
              wide_vit_slot(rc, V, shift, curOff, extraOff, m);
            }
          }
        }
        __syncthreads();                            // what the first wavefront stored is everybody's input again
      }
    }
    if (cells && (!P.lastOnly || c == outLen)) {
      double *col = P.lastOnly ? cells : cells + (long long)o * S;
      for (int k = tid; k < S; k += W) col[k] = V[curOff + k];
    }
    if (tid == 0) V[prevOff + S + 1] = -INFINITY;   // the seed is spent (this vector is the next column's `cur`)
    __syncthreads();                                // (the column is copied out and the seed cleared before the next column's first store)
    const int t = prevOff; prevOff = curOff; curOff = t;
  }
  if (loglike && tid == 0) loglike[bid] = V[prevOff + P.resultIdx];
}

// ---- retimed programs: a period of rounds, every node on its own column (see WideRetDev) ---------------------------------------
// lane-group reduction of a wavefront whose groups all have gWave lanes: no per-lane masks
template <int MODE, int H>
__device__ __forceinline__ void wide_max_all(double &m) {
  const double mo = __hiloint2double(wide_xor_lane<H>(__double2hiint(m)), wide_xor_lane<H>(__double2loint(m)));
  m = wide_max_raw(m, mo);
}
template <int H>
__device__ __forceinline__ void wide_sum_all(float &s) { s += __int_as_float(wide_xor_lane<H>(__float_as_int(s))); }
template <int MODE>
__device__ __forceinline__ void wide_group_reduce_all(double &m, float &s, int gWave) {
  const double own = m;
  wide_max_all<MODE, 1>(m);
  if (gWave > 2) wide_max_all<MODE, 2>(m);
  if (gWave > 4) wide_max_all<MODE, 4>(m);
  if (gWave > 8) wide_max_all<MODE, 8>(m);
  if (gWave > 16) wide_max_all<MODE, 16>(m);
  if (gWave > 32) wide_max_all<MODE, 32>(m);
  if (MODE == MB_VITERBI) return;
  s *= wide_exp_diff(own - m);
  wide_sum_all<1>(s);
  if (gWave > 2) wide_sum_all<2>(s);
  if (gWave > 4) wide_sum_all<4>(s);
  if (gWave > 8) wide_sum_all<8>(s);
  if (gWave > 16) wide_sum_all<16>(s);
  if (gWave > 32) wide_sum_all<32>(s);
}

// LDS by raw byte address (the records of the retimed sweep carry addresses, not indices)
typedef __attribute__((address_space(3))) double wide_lds_f64;
__device__ __forceinline__ double wide_lds_read(unsigned a) { return *(const wide_lds_f64 *)(uintptr_t)a; }
__device__ __forceinline__ void wide_lds_write(unsigned a, double v) { *(wide_lds_f64 *)(uintptr_t)a = v; }

// ---- the log-sum-exp correction term in fp64 (ACC variants of the retimed sum sweep: the fills of a one-tape E-step over long
// sequences).  With v_exp_f32 / v_log_f32 a state's per-column error is ~6e-8 -- and NOT random: a state that sits in a stationary
// regime (an intron's self-loop chain, the flanking states) sees nearly the same arguments column after column, so the same
// rounding repeats and the error grows LINEARLY, at a rate that differs between groups of states; the per-column normaliser of the
// count kernel removes only what a column's states share.  Measured at 50 000 columns under --use-defaults parameters: 7.6e-5 per
// transition against the exact oracle, whole groups of transitions sitting at that value (round 5).
// exp(x), x <= 0: 2^(k/64) from a 64-entry table in LDS times a degree-4 polynomial of the remainder (|r| <= ln 2 / 128: 4e-14).
__device__ __forceinline__ double wide_exp64(double x, unsigned tab) {
  x = wide_max_raw(x, -740.0);                                        // (-inf and the -1e300 stand-in: ~0)
  const double kf = __builtin_rint(x * 92.33248261689366);           // 64 / ln 2
  const double r = __builtin_fma(kf, -0.010830424696249145, x);      // x - k ln 2 / 64
  const int k = (int)kf;
  const double p = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(r, 0.041666666666666664, 0.16666666666666666), r, 0.5), r, 1.0), r, 1.0);
  return __builtin_ldexp(wide_lds_read(tab + ((unsigned)(k & 63) << 3)) * p, k >> 6);
}
// log(s), s >= 1: v_log_f32 as the seed, one Newton step through the exponential above (second order kept)
__device__ __forceinline__ double wide_log64(double s, unsigned tab) {
  const double t0 = (double)__log2f((float)s) * 0.6931471805599453;
  const double d = __builtin_fma(s, wide_exp64(-t0, tab), -1.0);
  return t0 + __builtin_fma(-0.5 * d, d, d);
}
template <int MODE>
__device__ __forceinline__ void wide_fold(double &m, double &s, double v, double sv, unsigned tab) {
  const double e = wide_exp64(-fabs(v - m), tab);
  const bool up = v > m;
  s = __builtin_fma(up ? s : sv, e, up ? sv : s);
  m = wide_max_raw(m, v);
}
template <int H>
__device__ __forceinline__ void wide_sum_step(double &s, int g) {
  const double so = __hiloint2double(wide_xor_lane<H>(__double2hiint(s)), wide_xor_lane<H>(__double2loint(s)));
  if (H < g) s += so;
}
template <int H>
__device__ __forceinline__ void wide_sum_all(double &s) { s += __hiloint2double(wide_xor_lane<H>(__double2hiint(s)), wide_xor_lane<H>(__double2loint(s))); }
template <int MODE>
__device__ __forceinline__ void wide_group_reduce(double &m, double &s, int g, int gWave, unsigned tab) {
  const double own = m;
  if (gWave > 1) wide_max_step<MODE, 1>(m, g);
  if (gWave > 2) wide_max_step<MODE, 2>(m, g);
  if (gWave > 4) wide_max_step<MODE, 4>(m, g);
  if (gWave > 8) wide_max_step<MODE, 8>(m, g);
  if (gWave > 16) wide_max_step<MODE, 16>(m, g);
  if (gWave > 32) wide_max_step<MODE, 32>(m, g);
  s *= wide_exp64(own - m, tab);
  if (gWave > 1) wide_sum_step<1>(s, g);
  if (gWave > 2) wide_sum_step<2>(s, g);
  if (gWave > 4) wide_sum_step<4>(s, g);
  if (gWave > 8) wide_sum_step<8>(s, g);
  if (gWave > 16) wide_sum_step<16>(s, g);
  if (gWave > 32) wide_sum_step<32>(s, g);
}
template <int MODE>
__device__ __forceinline__ void wide_group_reduce_all(double &m, double &s, int gWave, unsigned tab) {
  const double own = m;
  wide_max_all<MODE, 1>(m);
  if (gWave > 2) wide_max_all<MODE, 2>(m);
  if (gWave > 4) wide_max_all<MODE, 4>(m);
  if (gWave > 8) wide_max_all<MODE, 8>(m);
  if (gWave > 16) wide_max_all<MODE, 16>(m);
  if (gWave > 32) wide_max_all<MODE, 32>(m);
  s *= wide_exp64(own - m, tab);
  wide_sum_all<1>(s);
  if (gWave > 2) wide_sum_all<2>(s);
  if (gWave > 4) wide_sum_all<4>(s);
  if (gWave > 8) wide_sum_all<8>(s);
  if (gWave > 16) wide_sum_all<16>(s);
  if (gWave > 32) wide_sum_all<32>(s);
}

// GV: the ring lives in an L2-resident scratch vector of the workgroup (machines whose ring exceeds the LDS: the whole fn3
// profile composite, 21 761 states); penalties and the token window stay in LDS.  Records then carry ring ENTRIES (src >> 13).
// lane-group reduction of (value, place): the maximum, and among equal maxima the SMALLEST place -- the first maximum of the
// reference's enumeration (std::max_element, src/dpmatrix.defs.h:171-174); lanes of a group hold places slot * g + lane
// Two butterflies: the maximum first (the plain max reduction), then -- lanes that hold the maximum keep their place, the others take
// 0xFFFFFFFF -- the minimum of the places (one v_min_u32 per step instead of two fp64 compares and three selects on a pair).
template <int H>
__device__ __forceinline__ void wide_key_step(uint32_t &key, int g) {
  const uint32_t ko = (uint32_t)wide_xor_lane<H>((int)key);
  if (H < g) key = min(key, ko);
}
template <int H>
__device__ __forceinline__ void wide_rawmax_step(double &m, int g) {
  const double mo = __hiloint2double(wide_xor_lane<H>(__double2hiint(m)), wide_xor_lane<H>(__double2loint(m)));
  const double mx = wide_max_raw(m, mo);
  m = (H < g) ? mx : m;
}
__device__ __forceinline__ void wide_group_reduce_tb(double &m, uint32_t &key, int g, int gWave) {
  const double own = m;
  if (gWave > 1) wide_rawmax_step<1>(m, g);
  if (gWave > 2) wide_rawmax_step<2>(m, g);
  if (gWave > 4) wide_rawmax_step<4>(m, g);
  if (gWave > 8) wide_rawmax_step<8>(m, g);
  if (gWave > 16) wide_rawmax_step<16>(m, g);
  if (gWave > 32) wide_rawmax_step<32>(m, g);
  key = (own == m) ? key : 0xFFFFFFFFu;              // (a maximum is one of its operands, bit for bit; -inf == -inf: place 0 of the first lane wins)
  if (gWave > 1) wide_key_step<1>(key, g);
  if (gWave > 2) wide_key_step<2>(key, g);
  if (gWave > 4) wide_key_step<4>(key, g);
  if (gWave > 8) wide_key_step<8>(key, g);
  if (gWave > 16) wide_key_step<16>(key, g);
  if (gWave > 32) wide_key_step<32>(key, g);
}

// TB (max sweep only): `pool` holds one traceback CODE per cell (bytes, wide_tb_stride(S) per column, PairDesc::cellBase = byte
// offset) instead of the fp64 cell -- the place of the cell's first maximal candidate in its node's list (WideProgram::tbCodes)
// ACC (sum sweep only): the log-sum-exp correction term in fp64 (wide_exp64 / wide_log64 above) instead of v_exp_f32 / v_log_f32; the
// 64-entry table of 2^(j/64) sits in the last 512 bytes of the launch's LDS
// PART (k workgroups per sequence, see WidePartDev): `pp` is this workgroup's part, `A` the exchange buffer; the ring entries below
// pp->Sloc are the part's own states (matrix column pp->gmap[x]), imports enter through the penalty table, exports leave for A.X
constexpr unsigned long long WIDE_X_EMPTY = ~0ull;      // what the exchange buffer is preset to: a NaN no cell holds
__device__ __forceinline__ unsigned long long wide_x_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double wide_x_wait(const unsigned long long *p, unsigned *err, long long timeoutTicks) {
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    const unsigned long long b = wide_x_load(p);
    if (b != WIDE_X_EMPTY) return __longlong_as_double((long long)b);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return -INFINITY;      // some wait ran out: the call has failed, drain
    if ((long long)wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return -INFINITY; }
    __builtin_amdgcn_s_sleep(4);
  }
}

template <int MODE, bool GV, bool TB, bool ACC, bool PART, int RING, bool W2>
__device__ __forceinline__ void wide_retimed_body(const WideDev &P, const WideRetDev &Q, const PairDesc pd, const unsigned bid, const int *__restrict__ outTok,
                                                  double *__restrict__ pool, double *__restrict__ loglike, double *__restrict__ scratch,
                                                  const WidePartDev &part, const WidePartArgs &A) {
  static_assert(!(PART && GV), "the parts of a machine keep their rings in LDS");
  extern __shared__ double wlds[];
  // the records carry raw LDS byte addresses: the dynamic array is this kernel's only LDS, so it starts at 0
  if ((unsigned)(uintptr_t)wlds != 0u) __builtin_trap();
  const int tid = threadIdx.x, W = P.W;
  // (the part's fields in registers: loads from `part` behind the exchange's atomics would be issued again in every round epilogue)
  const int pSloc = part.Sloc, pImp = part.nImp, pExpBase = part.expBase, pExpIdx0 = part.expIdx0, pExp = part.nExp, pResult = part.resultEntry;
  const int S = PART ? pSloc : P.S, Sg = P.S;      // S: ring entries below it are cells of the matrix (S, S + 1: the constants); Sg: states of the machine
  const int L = P.inputTape ? pd.inLen : pd.outLen;
  const int NVs = Q.NVs, NB = Q.NB, nVec = NB * NVs, nPen = Q.nPen, rowLen = Q.rowLen;
  const int nImp = PART ? pImp : 0, nPenAll = nPen + nImp;
  double *V = GV ? scratch + (size_t)bid * (size_t)nVec : wlds;
  double *pen = GV ? wlds : wlds + nVec;             // [2][nPenAll]: this period's penalties (and imports) and the next one's
  int *tokWin = (int *)(pen + 2 * nPenAll);          // token of column c in entry c & 63, written two periods ahead
  const unsigned expTab = (unsigned)(((GV ? 0u : (unsigned)nVec * 8u) + 2u * (unsigned)nPenAll * 8u + (unsigned)WIDE_RET_TOKWIN * 4u + 7u) & ~7u);      // ACC: 2^(j/64), j = 0..63
  if (ACC && tid < 64) wide_lds_write(expTab + (unsigned)tid * 8u, exp2((double)tid * 0.015625));
  (void)expTab;
  const unsigned gmapOff = expTab + 512u;            // PART: machine state of every own state (32-bit)
  const size_t xStride = PART ? (size_t)A.nExpTot : 0;
  unsigned long long *xRow = PART ? (unsigned long long *)A.X + (size_t)A.xOff[bid] * xStride : nullptr;      // [column][exchange column] of this sequence
  const bool impLane = PART && tid >= W - nImp;      // the last lanes of the workgroup: one import each
  const int impI = tid - (W - nImp);
  const unsigned long long *impPtr = impLane ? xRow + part.impIdx[impI] : nullptr;
  unsigned long long impAhead = WIDE_X_EMPTY;
  if (PART && pool) for (int k = tid; k < S; k += W) *(unsigned *)((char *)wlds + gmapOff + 4u * (unsigned)k) = part.gmap[k];
  (void)gmapOff; (void)xRow; (void)impI; (void)impPtr; (void)impAhead; (void)Sg; (void)pExpBase; (void)pExpIdx0; (void)pExp; (void)pResult;
  for (int k = tid; k < nVec; k += W) V[k] = -INFINITY;
  if (tid < WIDE_RET_TOKWIN) tokWin[tid] = 0;
  __syncthreads();
  if (tid < NB) V[tid * NVs + S + 1] = 0.0;         // the seed's source, in every ring vector
  const int *out = outTok + (P.inputTape ? pd.inBase : pd.outBase);
  double *cells = (pool && !TB) ? pool + pd.cellBase : nullptr;
  unsigned char *codes = (pool && TB) ? (unsigned char *)pool + pd.cellBase : nullptr;
  const int Sb = (Sg + 3) & ~3;
  uint32_t bestSlot = 0u; int slotInRound = 0;                 // TB: slot of this lane's first maximal candidate within the round
  (void)codes; (void)Sb; (void)bestSlot; (void)slotInRound;
  auto tokAt = [&](int c) -> int { return (c >= 1 && c <= L) ? (P.backward ? out[L - c] : out[c - 1]) : 0; };
  if (tid == 0) tokWin[1] = tokAt(1);
  int tokNext = tid == 0 ? tokAt(2) : 0;
  // penalty of entry (ktau, col) in the period whose newest column is `newest`: col 0 silent, 1.. = the token, rowLen - 1 = the seed
  const int myKt = tid / rowLen, myCol = tid - myKt * rowLen;
  auto penalty = [&](int kt, int col, int newest) -> double {
    const int c = newest - kt;
    const bool ok = col == 0 || (col == rowLen - 1 ? c == 0 : (c >= 1 && tokWin[c & (WIDE_RET_TOKWIN - 1)] == col));
    return ok ? 0.0 : -INFINITY;
  };
  __syncthreads();
  for (int e = tid; e < nPen; e += W) { const int kt = e / rowLen; pen[e] = penalty(kt, e - kt * rowLen, 0); }
  if (impLane) {      // column 0 of the imports; column 1 is asked for now and looked at in the first period
    pen[nPen + impI] = wide_x_wait(impPtr, A.err, A.timeoutTicks);
    if (L >= 1) impAhead = wide_x_load(impPtr + xStride);
  }
  __syncthreads();
  // one record stream per rotation of the ring (the newest column sits in vector t mod NB); a stream runs on into the next one
  // (buffer loads: slot base in an SGPR offset, lane offset in one VGPR -- no 64-bit address arithmetic per slot)
  typedef __attribute__((ext_vector_type(4))) unsigned int rec_u32x4;
  const __amdgpu_buffer_rsrc_t recRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Q.rec, 0, 0x7fffffff, 0x00020000);
  // (W2 -- a part with two-transition candidates: the second weights are a stream of their own behind the records, [slot][lane] doubles)
  struct Rec2 { double w; uint32_t src, pad; double w2; };
  typedef typename std::conditional<W2, Rec2, WideRec>::type RecT;
  const int laneOff = tid * (int)sizeof(WideRec), slotBytes = W * (int)sizeof(WideRec), perStreamBytes = Q.nSlots * slotBytes;
  const int w2Base = W2 ? part.w2Offset + tid * 8 : 0;
  auto ldrec = [&](int soff) -> RecT {
    const rec_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(recRsrc, laneOff, soff, 0);
    RecT o; o.w = __hiloint2double((int)r.y, (int)r.x); o.src = r.z; o.pad = r.w;
    if constexpr (W2) {
      typedef __attribute__((ext_vector_type(2))) unsigned int rec_u32x2;
      const rec_u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(recRsrc, w2Base, soff >> 1, 0);      // (8 of the record's 16 bytes per lane and slot)
      o.w2 = __hiloint2double((int)r2.y, (int)r2.x);
    }
    return o;
  };
  RecT q[RING];
#pragma unroll
  for (int k = 0; k < RING; ++k) { q[k] = ldrec(k * slotBytes); __builtin_amdgcn_sched_barrier(0); }      // (in this order: the waits inside the loop count the loads behind a record)
  const int nPer = L + 1 + Q.kMax;
  unsigned penCur = GV ? 0u : (unsigned)nVec * 8u, penNxt = penCur + (unsigned)nPenAll * 8u;
  double m = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG;
  typename std::conditional<ACC, double, float>::type s = 0;
  const bool storeAll = cells && !P.lastOnly, storeLast = cells && P.lastOnly;
  // a node's lag comes as kq = kMax - ktau (ktau for a backward sweep): its column is cBase + cSign * kq, its matrix row starts
  // kq * rowS doubles behind rowPtr
  const int cSign = P.backward ? -1 : 1, rowS = P.lastOnly ? 0 : Sg;
  int cm = 0;
  auto ring = [&](uint32_t src) -> double { return GV ? *(const double *)((const char *)V + ((size_t)(src >> 13) << 3)) : wide_lds_read(src >> 14); };
  // the ring reads of a slot are issued one slot ahead (behind a barrier they are issued again: what they fetched may be stale)
  double vAhead = ring(q[0].src), pAhead = wide_lds_read(((q[0].src & 0x1fffu) << 3) + penCur);
  for (int t = 0; t < nPer; ++t) {
    if (tid == 0) {                                  // (the entry of column t - 62: no node lags that far)
      tokWin[(t + 2) & (WIDE_RET_TOKWIN - 1)] = tokNext;
      tokNext = tokAt(t + 3);
    }
    if (tid < nPen) wide_lds_write(penNxt + (unsigned)tid * 8u, penalty(myKt, myCol, t + 1));
    for (int e = tid + W; e < nPen; e += W) { const int kt = e / rowLen; wide_lds_write(penNxt + (unsigned)e * 8u, penalty(kt, e - kt * rowLen, t + 1)); }
    if (impLane) {      // the next period's newest column is t + 1: its value was asked for a period ago; the one after it is asked for now
      double v = -INFINITY;
      if (t + 1 <= L) v = impAhead != WIDE_X_EMPTY ? __longlong_as_double((long long)impAhead) : wide_x_wait(impPtr + (size_t)(t + 1) * xStride, A.err, A.timeoutTicks);
      wide_lds_write(penNxt + (unsigned)(nPen + impI) * 8u, v);
      if (t + 2 <= L) impAhead = wide_x_load(impPtr + (size_t)(t + 2) * xStride);
    }
    const int cBase = P.backward ? t : t - Q.kMax;
    const char *rowPtr = (const char *)(P.lastOnly ? cells : cells + (long long)(P.backward ? L - t : t - Q.kMax) * Sg);
    unsigned char *codeRow = TB ? codes + (long long)(t - Q.kMax) * Sb : nullptr;      // (forward sweeps only: column cBase = t - kMax)
    (void)codeRow;
    slotInRound = 0;      // (the padding slots behind the period's last round -- all -inf -- have been counted)
    const int streamBase = cm * perStreamBytes;
    for (int j0 = 0; j0 < Q.nSlots; j0 += RING) {
      // (the slot after this period's last one belongs to the next period: its penalties are the other table's)
      const unsigned penHere = penCur, penLast = j0 + RING == Q.nSlots ? penNxt : penCur;
#pragma unroll
      for (int k = 0; k < RING; ++k) {
        const RecT rc = q[k];
        const RecT &nx = q[(k + 1) % RING];          // the next slot's record (k = 7: the one requested at the end of this group's first slot)
        const unsigned penN = k + 1 == RING ? penLast : penHere;
        const double vNow = vAhead, pNow = pAhead;
        vAhead = ring(nx.src);
        pAhead = wide_lds_read(((nx.src & 0x1fffu) << 3) + penN);
        double cand = vNow + (rc.w + pNow);                    // w + 0.0 = w, w + -inf = -inf: the reference's one rounded add, or -inf
        if constexpr (W2) cand = cand + rc.w2;                 // (a two-transition candidate: the second transition's rounded add; + 0.0 otherwise)
        if (TB) { bestSlot = cand > m ? (uint32_t)slotInRound : bestSlot; ++slotInRound; }      // strict >: the first maximum
        if (MODE == MB_VITERBI) m = wide_max_raw(m, cand);
        else if constexpr (ACC) wide_fold<MODE>(m, s, cand, 1.0, expTab);
        else wide_fold<MODE>(m, s, cand, 1.0f);
        const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)rc.pad);
        if (flags & 0x80000000u) {
          const uint32_t dst = rc.pad;
          const int gWave = 1 << ((flags >> 26) & 7);
          uint32_t key = 0u;
          if (TB) {      // place in the node's list = slot * group + lane within the group (plan_stage deals candidate k to slot k / g, lane k % g)
            const int lg = (int)((dst >> 26) & 7u);
            key = (bestSlot << lg) | ((uint32_t)tid & ((1u << lg) - 1u));
            if (gWave > 1) wide_group_reduce_tb(m, key, 1 << lg, gWave);
            bestSlot = 0u; slotInRound = 0;
          } else
          if (gWave > 1) {
            if constexpr (ACC) {
              if (flags & 0x20000000u) wide_group_reduce<MODE>(m, s, 1 << ((dst >> 26) & 7), gWave, expTab);
              else wide_group_reduce_all<MODE>(m, s, gWave, expTab);
            } else {
              if (flags & 0x20000000u) wide_group_reduce<MODE>(m, s, 1 << ((dst >> 26) & 7), gWave);
              else wide_group_reduce_all<MODE>(m, s, gWave);
            }
          }
          const uint32_t x = dst & WIDE_RET_NO_DST, kq = (dst >> 20) & 63u;      // entry within its vector: the state (relays: >= S + 2)
          const int c = cSign < 0 ? cBase - (int)kq : cBase + (int)kq;
          if (x != WIDE_RET_NO_DST && (unsigned)c <= (unsigned)L) {      // (lanes without a node carry x = all ones)
            double res;
            if (MODE == MB_VITERBI) res = m;
            else if constexpr (ACC) res = s >= 0.5 ? m + (s == 1.0 ? 0.0 : wide_log64(s, expTab)) : -INFINITY;      // (a lone candidate: exact, as on the fp32 route)
            else res = s > 0.0f ? m + (double)__log2f(s) * 0.6931471805599453 : -INFINITY;      // (ln 2 in fp64: as a float it is 2.7e-9 too large, a bias that a column of hundreds of levels adds up)
            const uint32_t d = __umul24((dst >> 18) & 3u, (unsigned)NVs) + x;
            if (GV) V[d] = res; else wide_lds_write(d << 3, res);
            if constexpr (PART) {
              if (x - (unsigned)pExpBase < (unsigned)pExp) {      // an export (relay entries follow them): one 8-byte store that the consumers' lanes wait for (a NaN would read as "not yet": none is stored as all-ones)
                const unsigned long long bits = res == res ? (unsigned long long)__double_as_longlong(res) : 0x7ff8000000000000ull;
                __hip_atomic_store(xRow + (size_t)c * xStride + (size_t)(pExpIdx0 + (int)(x - (unsigned)pExpBase)), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              if (x < (unsigned)S) {
                if (TB) { if (codes) codeRow[__umul24(kq, (unsigned)Sb) + *(const unsigned *)((const char *)wlds + gmapOff + 4u * x)] = (unsigned char)key; }
                else if (storeAll | (storeLast & (c == L))) *(double *)(rowPtr + ((size_t)(__umul24(kq, (unsigned)rowS) + *(const unsigned *)((const char *)wlds + gmapOff + 4u * x)) << 3)) = res;
              }
            } else
            if (TB) { if (codes && x < (unsigned)S) codeRow[__umul24(kq, (unsigned)Sb) + x] = (unsigned char)key; }      // (traceback codes: forward sweeps only, cSign = 1)
            else if ((storeAll | (storeLast & (c == L))) && x < (unsigned)S) *(double *)(rowPtr + ((size_t)(__umul24(kq, (unsigned)rowS) + x) << 3)) = res;
          }
          m = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG;
          s = 0;
          if (flags & 0x40000000u) {
            __syncthreads();
            // both look-ups of the next slot again: its ring value may be stale, and so may its penalty -- the next period's table is
            // written at the top of THIS period, and when this barrier is the period's only one and sits in its last slot (a plain
            // HMM: period 1, slot count a multiple of RING) nothing but timing ordered those writes before the read above
            vAhead = ring(nx.src);
            pAhead = wide_lds_read(((nx.src & 0x1fffu) << 3) + penN);
          }
        }
        // the record of this slot one ring ahead, requested when the slot is done with its own: the load lands in the registers the old
        // record occupied (requested at the top of the slot it needs registers of its own, and the copies at the loop's back edge made
        // the compiler wait for ALL eight loads in flight there -- the prefetch ring was drained every eight slots)
        __builtin_amdgcn_sched_barrier(0);
        q[k] = ldrec(streamBase + (j0 + RING + k) * slotBytes);
      }
    }
    cm = cm + 1 == NB ? 0 : cm + 1;
    const unsigned sw = penCur; penCur = penNxt; penNxt = sw;
  }
  if (PART) { if (loglike && tid == 0 && pResult >= 0) loglike[bid] = V[(L % NB) * NVs + pResult]; }
  else if (loglike && tid == 0) loglike[bid] = V[(L % NB) * NVs + P.resultIdx];
}

template <int MODE, bool GV, bool TB = false, bool ACC = false>
__global__ __launch_bounds__(1024) void k_wide_retimed(WideDev P, WideRetDev Q, const PairDesc *__restrict__ pairs, const int *__restrict__ outTok,
                                                       double *__restrict__ pool, double *__restrict__ loglike, double *__restrict__ scratch) {
  wide_retimed_body<MODE, GV, TB, ACC, false, WIDE_RING, false>(P, Q, pairs[blockIdx.x], blockIdx.x, outTok, pool, loglike, scratch, WidePartDev{}, WidePartArgs{});
}

// first exchange row of every sequence (a handful of sequences: one lane)
__global__ void k_wide_part_rows(const PairDesc *__restrict__ pairs, int nSeq, int inputTape, long long *__restrict__ xOff) {
  long long rows = 0;
  for (int p = 0; p < nSeq; ++p) { xOff[p] = rows; rows += (long long)(inputTape ? pairs[p].inLen : pairs[p].outLen) + 1; }
}

// k workgroups per sequence: workgroup = part * nSeq + sequence (a part waits for lower parts only, and those are dispatched first)
template <int MODE, bool TB, bool ACC, bool W2>
__global__ __launch_bounds__(1024) void k_wide_retimed_parts(WideDev P, WidePartArgs A, const PairDesc *__restrict__ pairs, const int *__restrict__ outTok,
                                                             double *__restrict__ pool, double *__restrict__ loglike) {
  const unsigned part = blockIdx.x / (unsigned)A.nSeq, seq = blockIdx.x - part * (unsigned)A.nSeq;
  // the part's descriptor, in scalar registers (it is the same for every lane; the record stream's buffer descriptor is built from it)
  const WidePartDev g = A.parts[part];
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto uniPtr = [&](const void *q) { const unsigned long long b = (unsigned long long)(uintptr_t)q; return (const void *)(uintptr_t)(((unsigned long long)(unsigned)uni((int)(b >> 32)) << 32) | (unsigned)uni((int)b)); };
  WidePartDev pd;
  pd.ret.rec = (const WideRec *)uniPtr(g.ret.rec); pd.ret.nSlots = uni(g.ret.nSlots); pd.ret.NB = uni(g.ret.NB); pd.ret.NVs = uni(g.ret.NVs); pd.ret.kMax = uni(g.ret.kMax);
  pd.ret.rowLen = uni(g.ret.rowLen); pd.ret.nPen = uni(g.ret.nPen);
  pd.gmap = (const uint32_t *)uniPtr(g.gmap); pd.impIdx = (const uint32_t *)uniPtr(g.impIdx);
  pd.Sloc = uni(g.Sloc); pd.nImp = uni(g.nImp); pd.expBase = uni(g.expBase); pd.expIdx0 = uni(g.expIdx0); pd.nExp = uni(g.nExp); pd.resultEntry = uni(g.resultEntry);
  pd.w2Offset = uni(g.w2Offset);
  wide_retimed_body<MODE, false, TB, ACC, true, WIDE_RING, W2>(P, pd.ret, pairs[seq], seq, outTok, pool, loglike, nullptr, pd, A);
}

// ---- single precision relative to a per-column reference (log-sum-exp programs) ------------------------------------
static constexpr float W_NEG_BIG32 = -3.0e38f;

__device__ __forceinline__ void wide_fold32(float &m, float &s, float v, float sv) {
  const float e = __expf(-fabsf(v - m));
  const bool up = v > m;
  s = __fmaf_rn(up ? s : sv, e, up ? sv : s);
  m = fmaxf(m, v);
}
template <int H>
__device__ __forceinline__ float wide_xor_lane_f(float v) { return __int_as_float(wide_xor_lane<H>(__float_as_int(v))); }
template <int H>
__device__ __forceinline__ void wide_max_step32(float &m, int g) { const float mo = wide_xor_lane_f<H>(m); if (H < g) m = fmaxf(m, mo); }
__device__ __forceinline__ void wide_group_reduce32(float &m, float &s, int g, int gWave) {
  const float own = m;
  if (gWave > 1) wide_max_step32<1>(m, g);
  if (gWave > 2) wide_max_step32<2>(m, g);
  if (gWave > 4) wide_max_step32<4>(m, g);
  if (gWave > 8) wide_max_step32<8>(m, g);
  if (gWave > 16) wide_max_step32<16>(m, g);
  if (gWave > 32) wide_max_step32<32>(m, g);
  s *= __expf(own - m);
  if (gWave > 1) wide_sum_step<1>(s, g);
  if (gWave > 2) wide_sum_step<2>(s, g);
  if (gWave > 4) wide_sum_step<4>(s, g);
  if (gWave > 8) wide_sum_step<8>(s, g);
  if (gWave > 16) wide_sum_step<16>(s, g);
  if (gWave > 32) wide_sum_step<32>(s, g);
}

// HYB: the current column (and the extra entries) in LDS, the previous column in an L2-resident vector of the workgroup --
// for machines whose two columns do not fit the LDS even in single precision (the whole fn3 composition: 21 761 states).
// Only the emitting rounds read the previous column (slot flag WIDE_F_PREV, uniform), the column epilogue writes it.
template <bool GV, bool HYB>
__global__ __launch_bounds__(1024) void k_wide_sum32(WideDev32 P1, const PairDesc *__restrict__ pairs1, const int *__restrict__ outTok,
                                                     double *__restrict__ pool1, double *__restrict__ loglike1, float *__restrict__ scratch1,
                                                     WideDev32 P2, WideSecond X) {
  extern __shared__ float wldsf[];
  __shared__ float wmax[16];
  const bool second = blockIdx.x >= X.nFirst;     // fused launch: see k_wide_sweep
  const WideDev32 P = second ? P2 : P1;
  const PairDesc *pairs = second ? X.pairs : pairs1;
  double *pool = second ? X.pool : pool1, *loglike = second ? nullptr : loglike1;
  float *scratch = second ? (float *)X.scratch : scratch1;
  const unsigned bid = blockIdx.x - (second ? X.nFirst : 0u);
  const PairDesc pd = pairs[bid];
  const int tid = threadIdx.x, W = P.W, S = P.S, NV = P.NV;
  const int outLen = P.inputTape ? pd.inLen : pd.outLen, nA = P.nA, n = P.nA + P.nB;   // the one tape the machine has
  float *V = GV ? scratch + (size_t)bid * (size_t)(2 * NV + P.NX) : wldsf;
  float *Pg = HYB ? scratch + (size_t)bid * (size_t)NV : nullptr;
  for (int k = tid; k < (HYB ? NV + P.NX : 2 * NV + P.NX); k += W) V[k] = -INFINITY;
  if (HYB) for (int k = tid; k < NV; k += W) Pg[k] = -INFINITY;
  __syncthreads();
  if (tid == 0) (HYB ? Pg : V)[S + 1] = 0.0f;       // the seed, read by the first column only
  __syncthreads();
  const int extraOff = HYB ? NV : 2 * NV;
  const int *out = outTok + (P.inputTape ? pd.inBase : pd.outBase);
  double *cells = pool ? pool + pd.cellBase : nullptr;
  auto tokOf = [&](int c) -> int {
    if (c > outLen) return 0;
    const int o = P.backward ? outLen - c : c;
    return P.backward ? (o < outLen ? out[o] : 0) : (o ? out[o - 1] : 0);
  };
  const WideRec32 *cursor = P.segA + (size_t)tokOf(0) * P.strideA;
  int jn = 0, cc = 0, tokAhead = tokOf(1);
  WideRec32 q[WIDE_RING];
  auto fetch = [&](WideRec32 &dst) {
    dst = cursor[tid];
    cursor += W; ++jn;
    if (jn == nA) cursor = P.segB;
    if (jn == n) { jn = 0; ++cc; cursor = P.segA + (size_t)tokAhead * P.strideA; tokAhead = tokOf(cc + 1); }
  };
#pragma unroll
  for (int k = 0; k < WIDE_RING; ++k) fetch(q[k]);
  double R = 0.0;                                   // reference of the previous column: cell = R + entry
  unsigned long long fcur = P.flags[0];
  for (int c = 0; c <= outLen; ++c) {
    const int o = P.backward ? outLen - c : c;
    const int shift = HYB ? 0 : (c & 1) * 16;
    const int curOff = HYB ? 0 : ((c & 1) ? 0 : NV), prevOff = (c & 1) ? NV : 0;
    float m = W_NEG_BIG32, s = 0.0f, lm = -INFINITY;
    for (int j0 = 0; j0 < n; j0 += WIDE_RING) {
      const unsigned long long fnext = P.flags[(j0 + WIDE_RING < n ? j0 + WIDE_RING : 0) / WIDE_RING];
#pragma unroll
      for (int k = 0; k < WIDE_RING; ++k) {
        const unsigned fl = (unsigned)(fcur >> (8 * k)) & 0xffu;
        const WideRec32 rc = q[k];
        fetch(q[k]);
        if (!(fl & WIDE_F_CTRL)) {
          float x;
          if (HYB) x = (fl & WIDE_F_PREV) ? Pg[rc.src] : V[rc.src];
          else x = V[(rc.src >> shift) & 0xffffu];
          wide_fold32(m, s, x + rc.w, 1.0f);
        }
        if (fl & WIDE_F_END) {
          const uint32_t dst = q[(k + 1) % WIDE_RING].src;      // the control entry behind the round
          const int g = 1 << ((dst >> 26) & 7);
          const int gWave = 1 << (((uint32_t)__builtin_amdgcn_readfirstlane((int)dst) >> 26) & 7);
          if (gWave > 1) wide_group_reduce32(m, s, g, gWave);
          if ((dst & W_IDX_MASK) != W_NO_DST) {
            const float res = s > 0.0f ? __fmaf_rn(__log2f(s), 0.6931471805599453f, m) : -INFINITY;   // s in [1, lanes x slots]: v_log_f32 needs no range fix-up
            const bool extra = (dst >> 29) & 1;
            V[(extra ? extraOff : curOff) + (int)(dst & W_IDX_MASK)] = res;
            if (!extra) lm = fmaxf(lm, res);
          }
          m = W_NEG_BIG32; s = 0.0f;
          if (fl & WIDE_F_SYNC) __syncthreads();
        }
      }
      fcur = fnext;
    }
    // column maximum -> new reference; the column is rewritten relative to it (and copied out as fp64 cells)
#pragma unroll
    for (int h = 32; h; h >>= 1) lm = fmaxf(lm, __shfl_xor(lm, h, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = lm;
    __syncthreads();
    float M = -INFINITY;
    for (int w = 0; w < (W >> 6); ++w) M = fmaxf(M, wmax[w]);
    if (!(M > -INFINITY)) M = 0.0f;
    double *col = (cells && (!P.lastOnly || c == outLen)) ? (P.lastOnly ? cells : cells + (long long)o * S) : nullptr;
    for (int k = tid; k < S; k += W) {
      const float y = V[curOff + k];
      if (col) col[k] = R + (double)y;
      if (c == outLen && k == P.resultIdx && loglike) loglike[bid] = R + (double)y;   // the same rounding as the stored cell
      if (HYB) Pg[k] = y - M; else V[curOff + k] = y - M;
    }
    R += (double)M;
    if (tid == 0) { if (HYB) Pg[S + 1] = -INFINITY; else V[prevOff + S + 1] = -INFINITY; }   // the seed is spent
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// host: program compiler
// ------------------------------------------------------------------------------------------------------------
namespace {
struct WCand { uint32_t src; double w; uint32_t ref = 0xFFFFFFFFu; double w2 = 0.0; };      // ref: position of the transition in the machine's incoming view (levelled nodes); w2: second transition of a two-transition candidate (parts, see ret_merge)
struct WNode { uint32_t dst; int stage; std::vector<std::vector<WCand>> t2; std::vector<WCand> t3; };
inline uint32_t CUR(int i) { return (uint32_t)i; }
inline uint32_t EXTRA(int i) { return (1u << 30) | (uint32_t)i; }
inline uint32_t PREV(int i) { return (2u << 30) | (uint32_t)i; }

int env_int_w(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}

double host_lse2(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double mx = a > b ? a : b, mn = a > b ? b : a;
  return mx + log1p(exp(mn - mx));
}

int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// lane groups and rounds of one stage: returns the modelled cost (cycles), appends to P when `emit`
double plan_stage(const std::vector<const WNode *> &nodes, int nTokTables, int W, bool emit, WideProgram *P, double cRound = 105.0) {
  const int n = (int)nodes.size();
  if (!n) return 0.0;
  std::vector<int> len(n);
  int maxLen = 1;
  for (int i = 0; i < n; ++i) {
    int t2 = 0;
    for (const auto &l : nodes[i]->t2) t2 = std::max(t2, (int)l.size());
    len[i] = std::max(1, t2 + (int)nodes[i]->t3.size());
    maxLen = std::max(maxLen, len[i]);
  }
  // calibrated on the box (forced K sweeps, least squares over slots / rounds / barriers per column): a candidate slot of
  // 1024 lanes costs 0.19 us (fp64 columns in LDS) to 0.30 us (fp32, previous column in L2), a ROUND -- epilogue with the
  // lane-group reduction, log, store -- 0.30 to 0.58 us, i.e. 1.6-1.9 slots, and the barrier itself next to nothing
  // (the retimed kernel, whose slots are leaner: 0.08 us a slot, 0.37 us a round with its barrier -- cRound = 280)
  const double cSlot = 60.0, cShfl = 3.0, cSync = 10.0;
  const int maxGroup = std::max(1, std::min(64, env_int_w("MB_WIDE_MAX_GROUP", 64)));
  auto groupOf = [&](int L, int d) { return std::min(maxGroup, pow2ceil((L + d - 1) / d)); };
  int bestD = 1; double best = 1e300;
  for (int d = 1; d <= maxLen; ++d) {
    long long lanes = 0; int depth = 0, mg = 1;
    for (int i = 0; i < n; ++i) { const int g = groupOf(len[i], d); lanes += g; depth = std::max(depth, (len[i] + g - 1) / g); mg = std::max(mg, g); }
    const long long R = (lanes + W - 1) / W;
    const double c = R * (depth * cSlot + cRound) + cShfl * ilog2(mg) + cSync;
    if (c < best) { best = c; bestD = d; }
    if (R == 1 && depth <= d && d > 1 && mg == 1) break;
  }
  if (!emit) return best;
  // ---- emit the rounds ---------------------------------------------------------------------------------------
  std::vector<int> order(n), grp(n);
  for (int i = 0; i < n; ++i) grp[i] = groupOf(len[i], bestD);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return grp[a] != grp[b] ? grp[a] > grp[b] : len[a] > len[b]; });
  int pos = 0;
  while (pos < n) {
    int lanes = 0, e = pos, depth = 1, mg = 1;
    bool t2 = false;
    while (e < n && lanes + grp[order[e]] <= W) {
      const int i = order[e];
      lanes += grp[i]; depth = std::max(depth, (len[i] + grp[i] - 1) / grp[i]); mg = std::max(mg, grp[i]);
      for (const auto &l : nodes[i]->t2) if (!l.empty()) t2 = true;
      ++e;
    }
    WideRound R{};
    R.recBase = (int)P->recs.size(); R.depth = depth; R.dstBase = (int)P->dsts.size(); R.maxG = mg; R.sync = 0;
    R.pad0 = lanes;                                   // lanes that carry nodes
    const int nTab = t2 ? nTokTables : 1;
    R.tokStride = t2 ? depth * W : 0;
    const WideRec padRec{-INFINITY, PREV(P->dev.S), 0};
    P->recs.resize(P->recs.size() + (size_t)nTab * depth * W, padRec);
    if (P->wantW2) P->recs2.resize(P->recs.size(), 0.0);
    P->dsts.resize(P->dsts.size() + W, W_NO_DST);     // idle lanes: group of one, no destination
    int lane = 0;
    for (int q = pos; q < e; ++q) {
      const WNode &nd = *nodes[order[q]];
      const int g = grp[order[q]];
      for (int sub = 0; sub < g; ++sub)
        P->dsts[R.dstBase + lane + sub] = (sub == 0 ? ((nd.dst >> 30) << 29) | (nd.dst & W_IDX_MASK) : W_NO_DST) | ((uint32_t)ilog2(g) << 26);
      for (int t = 0; t < nTab; ++t) {
        const std::vector<WCand> *l2 = (t < (int)nd.t2.size()) ? &nd.t2[t] : nullptr;
        const int n2 = l2 ? (int)l2->size() : 0, tot = n2 + (int)nd.t3.size();
        for (int k = 0; k < tot; ++k) {       // (LDS bank conflicts of the data-dependent reads: measured negligible, SQ_LDS_BANK_CONFLICT)
          const WCand &cd = k < n2 ? (*l2)[k] : nd.t3[k - n2];
          const int j = k / g, sub = k % g;
          WideRec &rc = P->recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W + lane + sub];
          rc.w = cd.w; rc.src = cd.src;
          if (P->wantW2) P->recs2[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W + lane + sub] = cd.w2;
          if (t == nTab - 1) P->candsPerColumn++;
        }
      }
      lane += g;
    }
    P->slotsPerColumn += depth;
    P->rounds.push_back(R);
    pos = e;
  }
  P->rounds.back().sync = 1;
  P->nSync++;
  return best;
}
}  // namespace

bool wide_applicable(const mb_machine *m) {
  if ((m->nIn != 0) == (m->nOut != 0)) return false;      // exactly one tape: a generator (outputs only) or a recogniser (inputs only)
  if (!env_int_w("MB_WIDE", 1)) return false;
  return m->S >= env_int_w("MB_WIDE_MIN_STATES", 256) && m->S < (1 << 26);
}

void wide_free(WideProgram &P) {
  if (P.d_segA) (void)hipFree(P.d_segA);
  if (P.d_segB) (void)hipFree(P.d_segB);
  if (P.d_seg32A) (void)hipFree(P.d_seg32A);
  if (P.d_seg32B) (void)hipFree(P.d_seg32B);
  if (P.d_flags) (void)hipFree(P.d_flags);
  for (int k = 0; k < 4; ++k) if (P.d_vit[k]) (void)hipFree(P.d_vit[k]);
  if (P.d_phase) (void)hipFree(P.d_phase);
  if (P.d_ret) (void)hipFree(P.d_ret);
  if (P.d_tbOff) (void)hipFree(P.d_tbOff);
  if (P.d_tbEntry) (void)hipFree(P.d_tbEntry);
  for (WideJitKernel &J : P.jit) J.release();
  for (WidePartSet &ps : P.partSets) {
    for (WideJitKernel &J : ps.jit) J.release();
    for (WideRec *r : ps.d_rec) if (r) (void)hipFree(r);
    for (uint32_t *t : ps.d_tab) if (t) (void)hipFree(t);
    if (ps.d_parts) (void)hipFree(ps.d_parts);
    if (ps.d_tbOff) (void)hipFree(ps.d_tbOff);
    if (ps.d_tbEntry) (void)hipFree(ps.d_tbEntry);
  }
  P = WideProgram();
}

// rounds -> the two record streams the kernel reads (see WideDev)
static void wide_linearise(WideProgram &P, int nTok) {
  const int W = P.W, nR = (int)P.rounds.size();
  int lastTok = -1;
  for (int r = 0; r < nR; ++r) if (P.rounds[r].tokStride) lastTok = r;
  int nA = 0, nB = 0;
  for (int r = 0; r < nR; ++r) (r <= lastTok ? nA : nB) += P.rounds[r].depth;
  const int padSlots = (WIDE_RING - (nA + nB) % WIDE_RING) % WIDE_RING;
  (nB || lastTok < 0 ? nB : nA) += padSlots;
  const WideRec padRec{-INFINITY, PREV(P.dev.S), 0};
  P.segA.assign((size_t)nTok * nA * W, padRec);
  P.segB.assign((size_t)nB * W, padRec);
  for (int t = 0; t < nTok; ++t) {
    size_t ja = 0, jb = 0;
    for (int r = 0; r < nR; ++r) {
      const WideRound &R = P.rounds[r];
      const bool inA = r <= lastTok;
      if (!inA && t) continue;
      for (int j = 0; j < R.depth; ++j) {
        WideRec *dstp = inA ? &P.segA[((size_t)t * nA + ja++) * W] : &P.segB[(jb++) * W];
        const WideRec *srcp = &P.recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W];
        const bool last = j + 1 == R.depth;
        for (int l = 0; l < W; ++l) {
          dstp[l] = srcp[l];
          dstp[l].pad = last ? (0x80000000u | (R.sync ? 0x40000000u : 0u) | P.dsts[R.dstBase + l]) : 0u;
        }
      }
    }
  }
  P.dev.nA = nA; P.dev.nB = nB; P.dev.strideA = (long long)nA * W;
  // every entry of the workgroup's vectors addressable with 16 bits: both parities' indices precomputed in the record
  P.fastIdx = 2 * P.NV + P.NX <= 65536 && env_int_w("MB_WIDE_FAST_INDEX", 1);
  if (P.fastIdx) {
    const uint32_t NV = (uint32_t)P.NV;
    auto conv = [&](WideRec &rc) {
      const uint32_t sel = rc.src >> 30, idx = rc.src & 0x3fffffffu;
      const uint32_t even = sel == 0 ? NV + idx : (sel == 1 ? 2 * NV + idx : idx);      // even column: prev = vector 0, cur = vector 1
      const uint32_t odd = sel == 0 ? idx : (sel == 1 ? 2 * NV + idx : NV + idx);
      rc.src = even | (odd << 16);
    };
    for (WideRec &rc : P.segA) conv(rc);
    for (WideRec &rc : P.segB) conv(rc);
  }
}

template <class T>
static bool up_w(T *&d, const std::vector<T> &h);

// rounds -> the phase list and the four streams of k_wide_viterbi (see WideVitDev); needs P.segA / P.segB of wide_linearise
// only for their record format (16-bit indices for both column parities): the records are re-laid here from P.recs / P.dsts
static bool wide_vit_build(WideProgram &P, int nTok) {
  P.vitOk = false;
  if (!P.viterbi || !P.fastIdx || P.vecBytes() > WIDE_LDS_MAX || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0) || !env_int_w("MB_WIDE_VITERBI_PHASES", 1)) return true;
  const int W = P.W, nR = (int)P.rounds.size();
  int lastTok = -1;
  for (int r = 0; r < nR; ++r) if (P.rounds[r].tokStride) lastTok = r;
  const uint32_t NV = (uint32_t)P.NV;
  auto conv = [&](WideRec rc) {      // as wide_linearise: both parities' vector indices in the record
    const uint32_t sel = rc.src >> 30, idx = rc.src & 0x3fffffffu;
    const uint32_t even = sel == 0 ? NV + idx : (sel == 1 ? 2 * NV + idx : idx);
    const uint32_t odd = sel == 0 ? idx : (sel == 1 ? 2 * NV + idx : NV + idx);
    rc.src = even | (odd << 16);
    return rc;
  };
  const WideRec padRec = conv(WideRec{-INFINITY, PREV(P.dev.S), 0});
  std::vector<WidePhase> phases;
  std::vector<WideRec> st[4];        // wide A (one token table), wide B, thin A (one token table), thin B -- token tables appended below
  std::vector<std::vector<WideRec>> tokA[2];     // [wide / thin][token]
  tokA[0].assign(nTok, {}); tokA[1].assign(nTok, {});
  int r = 0;
  while (r < nR) {
    const bool thin = P.rounds[r].pad0 <= 64, inA = r <= lastTok;
    int e = r;
    while (e < nR && (P.rounds[e].pad0 <= 64) == thin && (e <= lastTok) == inA) ++e;
    const int lanes = thin ? 64 : W, ring = thin ? WIDE_THIN_RING : WIDE_RING;
    int slots = 0;
    for (int k = r; k < e; ++k) slots += P.rounds[k].depth;
    const int padded = (slots + ring - 1) / ring * ring;
    std::vector<WideRec> &one = inA ? tokA[thin ? 1 : 0][0] : st[thin ? 3 : 1];
    phases.push_back(WidePhase{thin ? 1 : 0, inA ? 1 : 0, padded, (int)(one.size() / lanes)});
    for (int t = 0; t < (inA ? nTok : 1); ++t) {
      std::vector<WideRec> &dstv = inA ? tokA[thin ? 1 : 0][t] : st[thin ? 3 : 1];
      for (int k = r; k < e; ++k) {
        const WideRound &R = P.rounds[k];
        for (int j = 0; j < R.depth; ++j) {
          const bool last = j + 1 == R.depth;
          // a barrier inside a thin phase is never needed (one wavefront); the phase itself ends with one
          const uint32_t uni = last ? (0x80000000u | ((R.sync && !thin) ? 0x40000000u : 0u)) : 0u;
          for (int l = 0; l < lanes; ++l) {
            WideRec rc = conv(P.recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W + l]);
            rc.pad = last ? (uni | P.dsts[R.dstBase + l]) : 0u;
            dstv.push_back(rc);
          }
        }
      }
      for (int k = slots; k < padded; ++k) for (int l = 0; l < lanes; ++l) dstv.push_back(padRec);
    }
    r = e;
  }
  if (phases.size() > 4096) return true;       // (a pathological alternation: the generic kernel keeps the machine)
  // a wide phase must end with a barrier before a thin phase reads its results: its last round closes a stage, which always
  // synchronises (only barriers between two thin rounds are ever dropped)
  P.vit.strideWideA = (long long)tokA[0][0].size(); P.vit.strideThinA = (long long)tokA[1][0].size();
  for (int t = 0; t < nTok; ++t) { st[0].insert(st[0].end(), tokA[0][t].begin(), tokA[0][t].end()); st[2].insert(st[2].end(), tokA[1][t].begin(), tokA[1][t].end()); }
  // slack behind every stream: the rings read one ring of slots past the end of a phase
  for (int k = 0; k < 4; ++k) st[k].insert(st[k].end(), (size_t)(k < 2 ? WIDE_RING * W : WIDE_THIN_RING * 64), padRec);
  for (int k = 0; k < 4; ++k) if (!up_w(P.d_vit[k], st[k])) return false;
  if (!up_w(P.d_phase, phases)) return false;
  P.vit.wideA = P.d_vit[0]; P.vit.wideB = P.d_vit[1]; P.vit.thinA = P.d_vit[2]; P.vit.thinB = P.d_vit[3];
  P.vit.phase = P.d_phase; P.vit.nPhases = (int)phases.size();
  P.vitOk = true;
  return true;
}

// the same rounds as 8-byte entries + control entries + slot flags for k_wide_sum32 (needs 16-bit vector indices)
static bool wide_linearise32(WideProgram &P, int nTok, bool hyb, std::vector<WideRec32> &segA, std::vector<WideRec32> &segB, std::vector<unsigned long long> &flagWords) {
  if (!hyb && 2 * P.NV + P.NX > 65536) return false;
  const int W = P.W, nR = (int)P.rounds.size();
  const uint32_t NV = (uint32_t)P.NV;
  int lastTok = -1;
  for (int r = 0; r < nR; ++r) if (P.rounds[r].tokStride) lastTok = r;
  int nA = 0, nB = 0;
  for (int r = 0; r < nR; ++r) (r <= lastTok ? nA : nB) += P.rounds[r].depth + 1;      // + the control entry
  const int padSlots = (WIDE_RING - (nA + nB) % WIDE_RING) % WIDE_RING;
  (nB || lastTok < 0 ? nB : nA) += padSlots;
  const int n = nA + nB;
  auto dual = [&](uint32_t src) {
    const uint32_t sel = src >> 30, idx = src & 0x3fffffffu;
    const uint32_t even = sel == 0 ? NV + idx : (sel == 1 ? 2 * NV + idx : idx);
    const uint32_t odd = sel == 0 ? idx : (sel == 1 ? 2 * NV + idx : NV + idx);
    return even | (odd << 16);
  };
  // HYB: one index per record -- into the previous column (slots flagged WIDE_F_PREV) or into the LDS image [cur | extra]
  const uint32_t ldsSentinel = NV + (uint32_t)P.NX - 1;        // the dummy extra entry: never written, always -inf
  auto isPad = [&](const WideRec &rc) { return rc.src == PREV(P.dev.S) && rc.w == -INFINITY; };
  const WideRec32 padRec{-INFINITY, hyb ? ldsSentinel : dual(PREV(P.dev.S))};
  segA.assign((size_t)nTok * nA * W, padRec);
  segB.assign((size_t)nB * W, padRec);
  std::vector<unsigned char> fl(n, WIDE_F_CTRL);       // padding slots are skipped
  // slot walk shared by the two passes: f(round, slot-in-round or depth for the control entry, token, slot in segment, global slot)
  auto walk = [&](auto &&f) {
    for (int t = 0; t < nTok; ++t) {
      size_t ja = 0, jb = 0;
      for (int r = 0; r < nR; ++r) {
        const bool inA = r <= lastTok;
        if (!inA && t) continue;
        for (int j = 0; j <= P.rounds[r].depth; ++j) {
          const size_t slot = inA ? ja++ : jb++;
          f(r, j, t, inA, slot, inA ? slot : nA + slot);
        }
      }
    }
  };
  std::vector<unsigned char> readsPrev(n, 0), readsCur(n, 0);
  if (hyb) {
    walk([&](int r, int j, int t, bool, size_t, size_t gslot) {
      const WideRound &R = P.rounds[r];
      if (j == R.depth) return;
      const WideRec *srcp = &P.recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W];
      for (int l = 0; l < W; ++l) if (!isPad(srcp[l])) ((srcp[l].src >> 30) == 2 ? readsPrev : readsCur)[gslot] = 1;
    });
    for (int j = 0; j < n; ++j) if (readsPrev[j] && readsCur[j]) return false;   // a slot reads one memory or the other (closure programs do)
  }
  walk([&](int r, int j, int t, bool inA, size_t slot, size_t gslot) {
    const WideRound &R = P.rounds[r];
    WideRec32 *dstp = inA ? &segA[((size_t)t * nA + slot) * W] : &segB[slot * W];
    if (j == R.depth) {                                  // control entry: the destination words of the round
      for (int l = 0; l < W; ++l) dstp[l] = WideRec32{-INFINITY, P.dsts[R.dstBase + l]};
      fl[gslot] = WIDE_F_CTRL;
      return;
    }
    const WideRec *srcp = &P.recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W];
    const unsigned char prevFlag = (hyb && readsPrev[gslot]) ? WIDE_F_PREV : 0;
    for (int l = 0; l < W; ++l) {
      uint32_t at;
      if (hyb) {
        const uint32_t sel = srcp[l].src >> 30, idx = srcp[l].src & 0x3fffffffu;
        at = isPad(srcp[l]) ? (prevFlag ? (uint32_t)P.dev.S : ldsSentinel) : (sel == 1 ? NV + idx : idx);
      } else at = dual(srcp[l].src);
      dstp[l] = WideRec32{(float)srcp[l].w, at};
    }
    // the last round of a column does not synchronise by itself: the kernel's column epilogue does
    fl[gslot] = (unsigned char)((j + 1 == R.depth ? (WIDE_F_END | ((R.sync && r + 1 < nR) ? WIDE_F_SYNC : 0)) : 0) | prevFlag);
  });
  flagWords.assign((size_t)n / WIDE_RING, 0ull);
  for (int j = 0; j < n; ++j) flagWords[j / WIDE_RING] |= (unsigned long long)fl[j] << (8 * (j % WIDE_RING));
  P.dev32.nA = nA; P.dev32.nB = nB; P.dev32.strideA = (long long)nA * W;
  return true;
}

// nodes of the program for K closure stages (K = 0: levelled)
static bool wide_nodes(const mb_machine *m, bool backward, int K, int W, long long pairCap, std::vector<WNode> &nodes, int &nExtra,
                       int &nStages, long long &nPairs) {
  const int S = m->S, nOut = m->nOut ? m->nOut : m->nIn;     // the alphabet of the machine's one tape
  const std::vector<int> &lev = backward ? m->levB : m->levF;
  const int nLev = backward ? m->nLevB : m->nLevF;
  const std::vector<int> &off = backward ? m->outOff : m->inOff;
  const std::vector<uint32_t> &perm = backward ? m->outPerm : m->inPerm;
  const int seedNode = backward ? S - 1 : 0;
  auto other = [&](uint32_t e) { return (int)(backward ? m->dst[e] : m->src[e]); };
  // candidates of a state: emitting (by output token) and silent, in the reference's iteration order
  std::vector<std::vector<std::vector<WCand>>> emitC(S, std::vector<std::vector<WCand>>(nOut + 1));
  struct SilE { int first; double second; uint32_t ref; };
  std::vector<std::vector<SilE>> sil(S);
  for (int x = 0; x < S; ++x)
    for (int tok = 0; tok <= nOut; ++tok) {
      const long long rw = (long long)x * (nOut + 1) + tok;      // row = (state * (nIn+1) + inTok) * (nOut+1) + outTok with one of the alphabets empty
      for (int a = off[rw]; a < off[rw + 1]; ++a) {
        const uint32_t e = perm[a];
        const int y = other(e);
        if (tok) emitC[x][tok].push_back({PREV(y), m->logW[e], (uint32_t)a});
        else if (backward ? y > x : y < x) sil[x].push_back({y, m->logW[e], (uint32_t)a});
      }
    }
  emitC[seedNode][0].push_back({PREV(S + 1), 0.0});
  nodes.clear(); nExtra = 0; nPairs = 0;
  if (K == 0) {
    for (int x = 0; x < S; ++x) {
      bool any = !sil[x].empty();
      for (const auto &l : emitC[x]) if (!l.empty()) any = true;
      if (!any) continue;
      WNode nd{CUR(x), lev[x], emitC[x], {}};
      for (auto &pe : sil[x]) nd.t3.push_back({CUR(pe.first), pe.second, pe.ref});
      nodes.push_back(std::move(nd));
    }
    nStages = nLev;
    return true;
  }
  // K > 0: the silent levels 1..nLev-1 cut into K groups of equal level count;  K < 0: ADAPTIVE -- levels are added to the
  // current stage while its closure stays within -K candidates, so the long thin runs of a profile's delete chain (a few
  // states per level, cheap to close) end up in few stages and the wide levels in stages of their own
  const bool adaptive = K < 0;
  const long long budget = adaptive ? -(long long)K : 0;
  if (!adaptive) K = std::max(1, std::min(K, std::max(1, nLev - 1)));
  std::vector<int> stg(S, 0);
  std::vector<char> isBase(S, 0);
  for (int x = 0; x < S; ++x) {
    if (!adaptive && lev[x] > 0) stg[x] = 1 + (int)(((long long)(lev[x] - 1) * K) / std::max(1, nLev - 1));
    for (const auto &l : emitC[x]) if (!l.empty()) isBase[x] = 1;
  }
  std::vector<int> eslot(S, -1);
  for (int x = 0; x < S; ++x)
    if (isBase[x] && !sil[x].empty()) eslot[x] = nExtra++;
  // closure: ancestors through silent paths whose intermediate states lie in the node's own stage, weights summed over paths
  std::vector<std::vector<std::pair<int, double>>> clos(S);
  std::vector<double> acc(S, -INFINITY);
  std::vector<int> touched;
  auto add = [&](int a, double w) {
    if (acc[a] == -INFINITY) { touched.push_back(a); acc[a] = w; }
    else acc[a] = host_lse2(acc[a], w);
  };
  auto closeRow = [&](int x) {          // uses stg[] of x and of everything before it in the sweep
    clos[x].clear();
    touched.clear();
    for (auto &pe : sil[x]) {
      const int y = pe.first; const double w = pe.second;
      if (w == -INFINITY) continue;
      if (stg[y] < stg[x]) { add(y, w); continue; }
      if (isBase[y]) add(y, w);
      for (auto &pa : clos[y]) add(pa.first, pa.second + w);
    }
    std::sort(touched.begin(), touched.end());
    clos[x].reserve(touched.size());
    for (int a : touched) { clos[x].push_back({a, acc[a]}); acc[a] = -INFINITY; }
    return (long long)clos[x].size();
  };
  if (!adaptive) {
    for (int q = 0; q < S; ++q) {
      const int x = backward ? S - 1 - q : q;
      if (sil[x].empty()) continue;
      nPairs += closeRow(x);
      if (nPairs > pairCap) return false;
    }
  } else {
    const std::vector<int> &levOff = backward ? m->levBOff : m->levFOff, &levState = backward ? m->levBState : m->levFState;
    auto closeLevel = [&](int L, int stage) {
      long long cands = 0;
      for (int k = levOff[L]; k < levOff[L + 1]; ++k) { const int x = levState[k]; stg[x] = stage; cands += closeRow(x) + (isBase[x] ? 1 : 0); }
      return cands;
    };
    std::vector<int> startOf;          // first level of every stage, ascending
    if (budget == WIDE_STAGES_DP) {
      // Optimal cut of the levels into stages under the planner's cost (slots of ~80 % filled lanes + one round per stage):
      // the rows of a stage depend on its first level only, so for every possible first level a the stage is grown level by
      // level once, and best[b] = min over a of best[a] + cost(levels a..b-1).
      const double cStage = 115.0, cSlotDp = 60.0, fill = 0.8 * W;
      const long long cap = 16ll * W;
      std::vector<double> best(nLev + 1, 1e300);
      std::vector<int> from(nLev + 1, -1);
      for (int x = 0; x < S; ++x) stg[x] = lev[x];          // every level its own stage: "earlier" == lower level
      best[1] = 0.0;
      for (int a = 1; a < nLev; ++a) {
        if (best[a] >= 1e300) continue;
        long long cum = 0; int L = a;
        for (; L < nLev; ++L) {
          cum += closeLevel(L, a);
          const double c = best[a] + std::ceil((double)cum / fill) * cSlotDp + cStage;
          if (c < best[L + 1]) { best[L + 1] = c; from[L + 1] = a; }
          if (cum > cap) { ++L; break; }
        }
        for (int l = a; l < L && l < nLev; ++l)
          for (int k = levOff[l]; k < levOff[l + 1]; ++k) stg[levState[k]] = l;
      }
      for (int b = nLev; b > 1; b = from[b]) { if (from[b] < 1) return false; startOf.push_back(from[b]); }
      std::reverse(startOf.begin(), startOf.end());
      int cur = 0; size_t nxt = 0;
      for (int L = 1; L < nLev; ++L) {
        if (nxt < startOf.size() && startOf[nxt] == L) { ++cur; ++nxt; }
        nPairs += closeLevel(L, cur);
        if (nPairs > pairCap) return false;
      }
      K = std::max(cur, 1);
    } else {
      int cur = 1; long long inStage = 0;
      for (int L = 1; L < nLev; ++L) {
        long long cands = closeLevel(L, cur);
        if (inStage > 0 && inStage + cands > budget) {       // the level opens a new stage: its rows are its direct predecessors
          ++cur; inStage = 0;
          cands = closeLevel(L, cur);
        }
        inStage += cands;
        nPairs += cands;
        if (nPairs > pairCap) return false;
      }
      K = cur;
    }
  }
  for (int x = 0; x < S; ++x) {
    if (isBase[x]) nodes.push_back(WNode{eslot[x] >= 0 ? EXTRA(eslot[x]) : CUR(x), 0, emitC[x], {}});
    if (!sil[x].empty()) {
      WNode nd{CUR(x), stg[x], {}, {}};
      if (isBase[x]) nd.t3.push_back({EXTRA(eslot[x]), 0.0});
      for (auto &pa : clos[x]) {
        const int a = pa.first;
        nd.t3.push_back({(stg[a] < stg[x] || eslot[a] < 0) ? CUR(a) : EXTRA(eslot[a]), pa.second});
      }
      if (!nd.t3.empty()) nodes.push_back(std::move(nd));
    }
  }
  nStages = K + 1;
  return true;
}

static double wide_plan(const std::vector<WNode> &nodes, int nStages, int nTok, int W, bool emit, WideProgram *P, double cRound = 105.0) {
  std::vector<std::vector<const WNode *>> byStage(nStages + 1);
  for (const WNode &n : nodes) byStage[std::min(n.stage, nStages)].push_back(&n);
  double cost = 0.0;
  for (auto &v : byStage) cost += plan_stage(v, nTok, W, emit, P, cRound);
  return cost;
}

template <class T>
static bool up_w(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(wide program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(wide program)")) return false;
  return true;
}

// ---- the retimed program of a levelled one-tape machine (see WideRetDev) --------------------------------------------------------
// One PART of a machine cut for k workgroups per sequence (see WidePartDev): the nodes handed to wide_ret_build are LOCAL -- vertices
// 0 .. nOwn - 1 the part's own states, then nImp import vertices (no candidates of their own: their value comes through the penalty
// table), then nExp export vertices (one silent candidate each: the exported state, weight 0.0); the seed's source is vertex count + 1
struct RetPart {
  int nOwn = 0, nImp = 0, nExp = 0, ring = WIDE_RING;
  bool useMerge = true;       // two-transition candidates (off: the part's LDS is tight -- they lengthen the spans, the relays push the period up)
  // the two-transition candidates of this part, chosen once (they depend on the graph, not on the lanes or the weights): indices into
  // the part's edge list in construction order, and that list's length as a check
  bool mergeKnown = false; std::vector<int> merged; size_t nEdges = 0;      // nEdges: ret_edge_sig of the list the indices refer to
};

namespace {
struct RetEdge { int src, dst, em, tok; double w; uint32_t ref; int rank; double w2 = 0.0; int orig = -1; };      // ref / rank: incoming-view position, place in the destination's candidate list (reference order); w2 / orig: see ret_merge

// smallest tau >= 0 with tau(dst) >= tau(src) + 1 - em * period over all edges; false when some tau would exceed `bound` (the period
// is shorter than a cycle of the machine needs, or the columns are deeper than the kernel's 6-bit lag)
bool ret_offsets(const std::vector<RetEdge> &edges, int nStates, int period, int bound, std::vector<int> &tau) {
  tau.assign(nStates, 0);
  long long work = 0;                                   // (a cap on the relaxation passes of one call: a pathological machine is refused, not waited for)
  for (;;) {
    bool moved = false;
    for (const RetEdge &e : edges) {
      const int t = tau[e.src] + 1 - (e.em ? period : 0);
      if (t > tau[e.dst]) { tau[e.dst] = t; moved = true; if (t > bound) return false; }
    }
    work += (long long)edges.size();
    if (!moved) return true;
    if (work > 400000000ll) return false;
  }
}

// the relaxation of ret_offsets at a period the edges do not allow: the edges (indices) of a cycle that gains time, empty when the
// offsets settle or only the depth bound was passed (predecessor pointers; their graph is checked for a cycle after every pass)
std::vector<int> ret_cycle(const std::vector<RetEdge> &edges, int nStates, int period, int bound) {
  std::vector<int> tau(nStates, 0), pred(nStates, -1), stamp(nStates, -1);
  for (int pass = 0; pass < 4 * nStates + 8; ++pass) {
    bool moved = false, over = false;
    for (int k = 0; k < (int)edges.size(); ++k) {
      const RetEdge &e = edges[k];
      const int t = tau[e.src] + 1 - (e.em ? period : 0);
      if (t > tau[e.dst]) { tau[e.dst] = t; pred[e.dst] = k; moved = true; over = over || t > bound; }
    }
    if (!moved) return {};
    for (int v0 = 0; v0 < nStates; ++v0) {
      if (stamp[v0] >= 0 || pred[v0] < 0) continue;
      int v = v0;
      while (v >= 0 && pred[v] >= 0 && stamp[v] < 0) { stamp[v] = v0; v = edges[pred[v]].src; }
      if (v >= 0 && pred[v] >= 0 && stamp[v] == v0) {      // walked into its own trail: a cycle through v
        std::vector<int> cyc;
        int u = v;
        do { cyc.push_back(pred[u]); u = edges[pred[u]].src; } while (u != v);
        std::reverse(cyc.begin(), cyc.end());
        return cyc;
      }
    }
    std::fill(stamp.begin(), stamp.end(), -1);
    if (over) return {};
  }
  return {};
}

// TWO-TRANSITION CANDIDATES (the parts of a machine cut for k workgroups per sequence).  A part's period is a chain of
// stages, one per transition of the machine's tightest cycle per emitted symbol (9 for the config-5 composite), and a stage costs the
// same whatever its width.  A silent transition v -> x is MERGED: x takes, instead of the candidate V(v) + w2, one candidate
// (V(u) + w1) + w2 per candidate u of v -- max over them = (max_u (V(u) + w1)) + w2 = V(v) + w2 bit for bit (rounding is monotone), a
// sum the same sum in another order -- so x no longer waits for v and the cycle is one stage shorter.  v keeps its own node (its cell
// is part of the matrix), a traceback code of x that falls into the merged block decodes to the transition v -> x, and the walker
// finds v's own code where it always was.  Merged: silent transitions only (an emitting second transition would need the source two
// columns back), never into a state that is itself read through a merge nor out of one (no three-transition candidates), not out
// of the seed's state, fan-ins bounded.  Chosen greedily along the cycles that forbid the next shorter period, until `target`.
// signature of a part's edge list (source, destination, emitting, token of every edge, in order, and their number): the merges chosen
// for a part are indices into this list, so they are re-applied only to the very same list -- a weight update that turns an edge to
// -inf and brings another back keeps the COUNT but not the edges
static size_t ret_edge_sig(const std::vector<RetEdge> &edges, int nStates, int nOwn) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) { h ^= v; h *= 1099511628211ull; };
  mix(edges.size()); mix((unsigned long long)nStates); mix((unsigned long long)nOwn);
  for (const RetEdge &e : edges) { mix(((unsigned long long)(unsigned)e.src << 32) | (unsigned)e.dst); mix(((unsigned long long)(unsigned)e.em << 32) | (unsigned)e.tok); }
  return (size_t)h;
}

void ret_merge(std::vector<RetEdge> &edges, int nStates, int nOwn, int seedState, int target, int bound, bool backward, bool verbose, RetPart *cache) {
  const size_t sig = ret_edge_sig(edges, nStates, nOwn);
  std::vector<RetEdge> orig = edges;
  for (size_t i = 0; i < orig.size(); ++i) orig[i].orig = (int)i;
  edges = orig;
  std::vector<char> merged(orig.size(), 0), isSrc(nStates, 0), isTgt(nStates, 0);
  std::vector<std::vector<int>> inOf(nStates);
  std::vector<int> fanIn(nStates, 0);
  for (size_t i = 0; i < orig.size(); ++i) { inOf[orig[i].dst].push_back((int)i); fanIn[orig[i].dst]++; }
  for (auto &l : inOf) std::stable_sort(l.begin(), l.end(), [&](int a, int b) { return orig[a].rank < orig[b].rank; });
  auto order = [&](std::vector<RetEdge> &ev) {
    std::stable_sort(ev.begin(), ev.end(), [&](const RetEdge &a, const RetEdge &b) { if (a.em != b.em) return a.em < b.em; return backward ? a.dst > b.dst : a.dst < b.dst; });
  };
  auto rebuild = [&]() {
    edges.clear();
    for (size_t i = 0; i < orig.size(); ++i) {
      if (!merged[i]) { edges.push_back(orig[i]); continue; }
      for (int f : inOf[orig[i].src]) edges.push_back(RetEdge{orig[f].src, orig[i].dst, orig[f].em, orig[f].tok, orig[f].w, orig[i].ref, orig[i].rank, orig[i].w, -1});
    }
    order(edges);
  };
  std::vector<int> tau;
  auto pMinOf = [&]() {
    int lo = 1, hi = 64;
    if (!ret_offsets(edges, nStates, hi, bound + 64, tau)) return 65;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (ret_offsets(edges, nStates, mid, 62 * mid + mid - 1, tau)) hi = mid; else lo = mid + 1; }
    return lo;
  };
  int nMerged = 0, p0 = -1, p = -1;
  if (cache && cache->mergeKnown && cache->nEdges == sig) {      // chosen before for this very edge list (another lane count, a weight update): apply
    for (int i : cache->merged) if (i >= 0 && i < (int)orig.size()) merged[i] = 1;
    if (!cache->merged.empty()) rebuild();
    return;
  }
  for (int iter = 0; iter < 4000; ++iter) {
    p = pMinOf();
    if (p0 < 0) p0 = p;
    if (p <= target || p > 64) break;
    const std::vector<int> cyc = ret_cycle(edges, nStates, p - 1, 62 * (p - 1) + p - 2);
    if (verbose && opt_env("MB_WIDE_VERBOSE_MERGE")) fprintf(stderr, "[mbhip]   merge iteration %d: period %d, cycle of %zu edges\n", iter, p, cyc.size());
    if (cyc.empty()) break;
    int picked = 0;
    for (int k : cyc) {
      const RetEdge &ce = edges[k];
      if (ce.orig < 0 || ce.em || merged[ce.orig]) continue;
      const int v = ce.src, x = ce.dst;
      if (v >= nOwn || x >= nOwn || v == seedState || isTgt[v] || isSrc[x] || inOf[v].empty()) continue;
      if ((int)inOf[v].size() > 8 || fanIn[x] + (int)inOf[v].size() - 1 > 48) continue;
      merged[ce.orig] = 1; isSrc[v] = 1; isTgt[x] = 1; fanIn[x] += (int)inOf[v].size() - 1;
      ++picked; ++nMerged;
    }
    if (!picked) break;
    rebuild();
  }
  if (cache) { cache->mergeKnown = true; cache->nEdges = sig; cache->merged.clear(); for (size_t i = 0; i < orig.size(); ++i) if (merged[i]) cache->merged.push_back((int)i); }
  if (verbose) fprintf(stderr, "[mbhip] wide retimed part: %d silent transitions merged into two-transition candidates, shortest period %d -> %d (%zu -> %zu candidates)\n", nMerged, p0, p, orig.size(), edges.size());
}
}  // namespace

static bool g_quiet_search = false;      // the lane search of wide_parts_host plans every candidate: its builds stay silent under MB_WIDE_VERBOSE
// hostOut: keep the record stream on the host instead of uploading it (mb_debug_wide_retimed: the planner without a device)
static bool wide_ret_build(const mb_machine *m, WideProgram &P, const std::vector<WNode> &nodes, int nTok, std::vector<WideRec> *hostOut = nullptr,
                           int keepPeriod = 0, const RetPart *part = nullptr) {
  P.retOk = false;
  const int W = P.W;
  // S: vertices of the retiming graph; SC: ring entry of the -inf constant (entries below it are the cells the matrix keeps)
  const int S = part ? part->nOwn + part->nImp + part->nExp : m->S, SC = part ? part->nOwn : m->S;
  const int nImp = part ? part->nImp : 0;
  auto entryOf = [&](int v) { return v < SC ? v : v + 2; };      // (the two constants sit between the own states and the other vertices)
  const int want = env_int_w("MB_WIDE_RETIMED", 1);
  if (want == 0) return true;
  const int rowLen = nTok + 1;                                 // penalty columns: silent, tokens 1 .. nTok - 1, the seed
  if (S + 2 >= (int)WIDE_RET_NO_DST || rowLen > 64) return true;
  if (part && (hostOut == nullptr || nImp > W)) return true;
  const bool verbose = opt_env("MB_WIDE_VERBOSE") != nullptr && !g_quiet_search;
  // the levelled nodes as a graph over states: t2[tok] = emitting candidates (source in the column before), t3 = silent ones
  std::vector<char> live(S, 0);
  for (const WNode &nd : nodes) live[nd.dst & W_IDX_MASK] = 1;
  std::vector<RetEdge> edges;
  int seedState = -1; double seedW = 0.0;
  for (const WNode &nd : nodes) {
    const int x = (int)(nd.dst & W_IDX_MASK);
    int rank = 0;      // the reference's traceback enumerates a cell's candidates emitting first, then silent, each in `incoming` order (src/dpmatrix.defs.h:93-103)
    for (int tok = 0; tok < (int)nd.t2.size(); ++tok)
      for (const WCand &cd : nd.t2[tok]) {
        const int y = (int)(cd.src & 0x3fffffffu);
        if (y == S + 1) { seedState = x; seedW = cd.w; continue; }      // the seed: no dependency
        if (y >= S || tok == 0) return true;                           // (cannot happen: token 0 of a levelled node holds the seed only)
        if (live[y] && cd.w != -INFINITY) edges.push_back(RetEdge{y, x, 1, tok, cd.w, cd.ref, rank});
        ++rank;
      }
    for (const WCand &cd : nd.t3) {
      const int y = (int)(cd.src & 0x3fffffffu);
      if (live[y] && cd.w != -INFINITY) edges.push_back(RetEdge{y, x, 0, 0, cd.w, cd.ref, rank});
      ++rank;
    }
  }
  // silent edges in sweep order first, so that one relaxation pass carries a change through a whole column
  std::stable_sort(edges.begin(), edges.end(), [&](const RetEdge &a, const RetEdge &b) {
    if (a.em != b.em) return a.em < b.em;
    return P.backward ? a.dst > b.dst : a.dst < b.dst;
  });
  const int kLimit = WIDE_RET_TOKWIN - 2;                      // largest lag the token window serves
  // parts: two-transition candidates until the period is about half the machine's own (MB_ONETAPE_PART_MERGE=0: none; n > 1: until period n)
  const int mergeTo = (part && part->useMerge) ? env_int_w("MB_ONETAPE_PART_MERGE", 1) : 0;
  if (mergeTo > 0) {
    int target = mergeTo;
    if (mergeTo == 1 && !part->mergeKnown) {
      std::vector<int> t0; int lo = 1, hi = 64;
      if (ret_offsets(edges, S, hi, kLimit * hi + hi - 1, t0)) { while (lo < hi) { const int mid = (lo + hi) / 2; if (ret_offsets(edges, S, mid, kLimit * mid + mid - 1, t0)) hi = mid; else lo = mid + 1; } target = (lo + 1) / 2; }
    }
    if (target >= 1) ret_merge(edges, S, SC, seedState, target, kLimit * 64 + 63, P.backward, verbose, const_cast<RetPart *>(part));
  }
  // the order candidates enter their node's list: the relaxation's order, or -- traceback codes -- the reference's enumeration order
  std::vector<RetEdge> byRank;
  const std::vector<RetEdge> *edgeOrder = &edges;
  if (P.tbCodes) {
    byRank = edges;
    std::stable_sort(byRank.begin(), byRank.end(), [](const RetEdge &a, const RetEdge &b) { return a.dst != b.dst ? a.dst < b.dst : a.rank < b.rank; });
    edgeOrder = &byRank;
  }
  std::vector<int> tau;
  auto feasible = [&](int period) { return ret_offsets(edges, S, period, kLimit * period + period - 1, tau); };
  // (a weight refresh keeps the period that was chosen: the schedule depends on the machine's structure, not on its weights)
  const int forced = keepPeriod > 0 ? keepPeriod : env_int_w("MB_WIDE_RETIMED_PERIOD", 0);
  int lo = 1, hi = 64;
  if (keepPeriod <= 0) {
    if (!feasible(hi)) return true;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (feasible(mid)) hi = mid; else lo = mid + 1; }
  }
  const int pMin = keepPeriod > 0 ? P.retPeriodMin : lo;
  // shape for one period length: ring depth, relays, nodes by residue; cost from the round planner
  struct Shape { int period = 0, NB = 0, NVs = 0, nRelay = 0, kMax = 0, tauMax = 0; bool gv = false; double cost = 1e300; std::vector<WNode> nd; };
  const bool forceGv = env_int_w("MB_WIDE_GLOBAL_VECTORS", 0) != 0, allowGv = env_int_w("MB_WIDE_RETIMED_L2", 1) != 0;
  auto shape = [&](int period, Shape &sh) -> bool {
    if (!feasible(period)) return false;
    int tauMax = 0;
    for (int x = 0; x < S; ++x) if (live[x]) tauMax = std::max(tauMax, tau[x]);
    // the deepest ring that fits the LDS (fewest relays); when none does, a ring of 4 in L2
    for (int pass = 0; pass < 2; ++pass)
    for (int NB = 4; NB >= 2; --NB) {
      // a value is readable for NB * period - 1 steps after it was written; sources with later readers are copied every `hop`
      // steps into relay entries of their own (relay k of u: a silent copy of relay k - 1 at tau(u) + k * hop)
      const int hop = NB * period - 1;
      std::vector<int> nHops(S, 0), relayBase(S, 0);
      for (const RetEdge &e : edges) {
        const int span = tau[e.dst] + (e.em ? period : 0) - tau[e.src];
        nHops[e.src] = std::max(nHops[e.src], (span - 1) / hop);
      }
      int nRelay = 0, kMax = 0;
      for (int x = 0; x < S; ++x) { relayBase[x] = nRelay; nRelay += nHops[x]; kMax = std::max(kMax, (tau[x] + nHops[x] * hop) / period); }
      const int NVs = S + 2 + nRelay, nPen = (kMax + 1) * rowLen;
      if (kMax > kLimit) continue;
      const size_t ldsPen = 2 * (size_t)(nPen + nImp) * sizeof(double) + WIDE_RET_TOKWIN * sizeof(int) + (part ? 512 + 4 * (size_t)SC : 0),
                   ldsAll = ldsPen + (size_t)NB * NVs * sizeof(double);
      const bool inLds = ldsAll <= WIDE_LDS_MAX && !forceGv;
      if ((pass == 0) != inLds) continue;
      if (part && !inLds) continue;                            // (parts keep their ring in LDS: that is what they are cut for)
      if ((!inLds && (ldsPen > WIDE_LDS_MAX || !allowGv)) || (size_t)NB * NVs >= (1u << 19) || NVs >= (int)WIDE_RET_NO_DST || nPen + nImp > 0x2000) continue;
      auto srcWord = [&](int ktDst, int em, int col, int entry) {      // penalty entry << 19 | ring entry for rotation 0
        const int back = (ktDst + em) % NB;
        return ((uint32_t)(ktDst * rowLen + col) << 19) | (uint32_t)(((NB - back) % NB) * NVs + entry);
      };
      auto dstWord = [&](int kt, int entry) { return ((uint32_t)kt << 20) | ((uint32_t)((NB - kt % NB) % NB) << 18) | (uint32_t)entry; };
      std::vector<std::vector<WCand>> cands(S);
      if (seedState >= 0 && !P.tbCodes) cands[seedState].push_back(WCand{srcWord(tau[seedState] / period, 0, rowLen - 1, SC + 1), seedW});
      for (const RetEdge &e : *edgeOrder) {
        const int kt = tau[e.dst] / period, span = tau[e.dst] + (e.em ? period : 0) - tau[e.src];
        const int k = (span - 1) / hop;
        cands[e.dst].push_back(WCand{srcWord(kt, e.em, e.tok, k ? S + 2 + relayBase[e.src] + k - 1 : entryOf(e.src)), e.w, e.ref, e.w2});
      }
      // (traceback codes: a cell's code is the PLACE of its first maximal candidate in this list -- the reference's order, the seed last)
      if (seedState >= 0 && P.tbCodes) cands[seedState].push_back(WCand{srcWord(tau[seedState] / period, 0, rowLen - 1, SC + 1), seedW});
      // an import: 0.0 (the constant) + (0.0 + its entry of the penalty table, behind the (ktau, token) entries) -- nothing leads to an
      // import vertex, so it sits in the period's newest column (ktau 0), which is the column its entry is loaded for
      bool importsNewest = true;
      for (int i = 0; i < nImp; ++i) {
        importsNewest = importsNewest && tau[SC + i] < period;
        cands[SC + i].push_back(WCand{((uint32_t)(nPen + i) << 19) | (uint32_t)(SC + 1), 0.0});
      }
      if (!importsNewest) continue;
      sh.nd.clear();
      for (int x = 0; x < S; ++x) {      // (states nothing leads to are nodes too: their cells of the matrix are -inf)
        sh.nd.push_back(WNode{CUR(dstWord(tau[x] / period, entryOf(x))), tau[x] % period, {}, std::move(cands[x])});
        for (int k = 1; k <= nHops[x]; ++k) {
          const int tr = tau[x] + k * hop, kr = tr / period, entry = S + 2 + relayBase[x] + k - 1;
          sh.nd.push_back(WNode{CUR(dstWord(kr, entry)), tr % period, {}, {WCand{srcWord(kr, 0, 0, k == 1 ? entryOf(x) : entry - 1), 0.0}}});
        }
      }
      sh.period = period; sh.NB = NB; sh.NVs = NVs; sh.nRelay = nRelay; sh.kMax = kMax; sh.tauMax = tauMax; sh.gv = !inLds;
      sh.cost = wide_plan(sh.nd, period - 1, 1, W, false, nullptr, 280.0);
      return true;
    }
    return false;
  };
  Shape best, cur;
  for (int period = forced > 0 ? forced : pMin; period <= (forced > 0 ? forced : std::min(64, pMin + 8)); ++period) {
    if (!shape(period, cur)) continue;
    if (verbose) fprintf(stderr, "[mbhip] wide retimed, period %d: tauMax %d, ring %d x %d (%d relays), modelled %.0f cycles per column\n",
                         period, cur.tauMax, cur.NB, cur.NVs, cur.nRelay, cur.cost);
    if (cur.cost < best.cost) std::swap(best, cur);
  }
  if (best.cost >= 1e300) return true;
  // rounds (one stage per residue) -> one stream of [slot][lane] records
  WideProgram T;
  T.W = W; T.dev.S = SC; T.wantW2 = part && part->useMerge;
  wide_plan(best.nd, best.period - 1, 1, W, true, &T, 280.0);
  if (T.rounds.empty()) return true;
  int nSlots = 0;
  for (const WideRound &R : T.rounds) nSlots += R.depth;
  // (the kernel keeps `ring` records in flight and takes its slots in groups of that many)
  const int ring = part ? part->ring : WIDE_RING;
  const int padded = (nSlots + ring - 1) / ring * ring;
  const bool withW2 = part && part->useMerge;                  // a second stream behind the records: the second weights, [slot][lane] doubles
  if ((size_t)(best.NB * padded + WIDE_RING) * W * sizeof(WideRec) * (withW2 ? 2 : 1) >= ((size_t)1 << 31)) return true;      // (the kernel's buffer loads carry 32-bit offsets)
  // one stream per rotation cm of the ring (newest column in vector cm): a record names its source by LDS byte address
  //   src = byte address << 14 | penalty entry;   pad (last slot) = flags | kq << 20 | vector << 18 | ring entry (see WideRetDev)
  const int NB = best.NB, NVs = best.NVs;
  std::vector<WideRec> st;
  st.reserve((size_t)(NB * padded + WIDE_RING) * W * (withW2 ? 2 : 1));
  std::vector<double> st2;
  for (int cm = 0; cm < NB; ++cm) {
    const bool gv = best.gv;                                   // records name ring entries (L2 ring) or LDS byte addresses
    const WideRec padRec{-INFINITY, gv ? (uint32_t)SC << 13 : (uint32_t)(SC * 8) << 14, 0};      // entry S of vector 0 (-inf), penalty entry 0 (0.0)
    const size_t start = st.size();
    for (const WideRound &R : T.rounds)
      for (int j = 0; j < R.depth; ++j) {
        const bool last = j + 1 == R.depth;
        for (int l = 0; l < W; ++l) {
          WideRec rc = T.recs[(size_t)R.recBase + (size_t)j * W + l];
          const double w2 = (withW2 && rc.w != -INFINITY) ? T.recs2[(size_t)R.recBase + (size_t)j * W + l] : 0.0;
          if (rc.w == -INFINITY) rc = padRec;                    // (the planner's own padding)
          else {
            const uint32_t a0 = rc.src & 0x7ffffu, penIdx = rc.src >> 19;      // shape(): entry for rotation 0, penalty entry
            const uint32_t vec = (a0 / NVs + cm) % NB, entry = vec * NVs + a0 % NVs;
            rc.src = (gv ? entry << 13 : (entry * 8u) << 14) | penIdx;
          }
          rc.pad = 0u;
          if (last) {
            const uint32_t dw = T.dsts[R.dstBase + l], x = dw & WIDE_RET_NO_DST, kt = (dw >> 20) & 63u, nkm = (dw >> 18) & 3u;
            uint32_t word = dw & 0x1c000000u;                    // log2 of the lane group
            if (x == WIDE_RET_NO_DST) word |= WIDE_RET_NO_DST;
            else word |= ((P.backward ? kt : (uint32_t)best.kMax - kt) << 20) | (((nkm + cm) % NB) << 18) | x;
            rc.pad = 0x80000000u | (R.sync ? 0x40000000u : 0u) | word;
            if (l % 64 == 0) {                                   // groups of different sizes in this wavefront: masked reduction
              const uint32_t g0 = (dw >> 26) & 7u;
              for (int q = l; q < std::min(W, l + 64); ++q) {
                const uint32_t dq = T.dsts[R.dstBase + q];
                if ((dq & WIDE_RET_NO_DST) != WIDE_RET_NO_DST && ((dq >> 26) & 7u) != g0) rc.pad |= 0x20000000u;
              }
            }
          }
          st.push_back(rc);
          if (withW2) st2.push_back(w2);
        }
      }
    st.resize(start + (size_t)padded * W, padRec);
    if (withW2) st2.resize(st.size(), 0.0);
  }
  st.insert(st.end(), st.begin(), st.begin() + (size_t)WIDE_RING * W);      // the ring reads one ring of slots into the next period
  if (withW2) {      // ... and the second weights behind them (two to a WideRec), read at half the records' byte offset
    st2.insert(st2.end(), st2.begin(), st2.begin() + (size_t)WIDE_RING * W);
    P.retW2Offset = (long long)st.size() * (long long)sizeof(WideRec);
    for (size_t q = 0; q < st2.size(); q += 2) { WideRec two; two.w = st2[q]; const double hi = q + 1 < st2.size() ? st2[q + 1] : 0.0; std::memcpy(&two.src, &hi, 8); st.push_back(two); }
  } else P.retW2Offset = 0;
  if (verbose)
    for (const WideRound &R : T.rounds) {
      int hist[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < W; ++l) { const uint32_t dw = T.dsts[R.dstBase + l]; if ((dw & WIDE_RET_NO_DST) != WIDE_RET_NO_DST) hist[(dw >> 26) & 7u]++; }
      fprintf(stderr, "[mbhip]   round: %d lanes, depth %d, sync %d, nodes by group size 1/2/4/8/16/32/64: %d %d %d %d %d %d %d\n", R.pad0, R.depth, R.sync, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6]);
    }
  if (hostOut) hostOut->swap(st);
  else {
    if (!up_w(P.d_ret, st)) return false;
    if (!best.gv && !part) P.h_ret.swap(st);      // (kept for the generated kernel: its per-lane table is built from the streams on first use)
  }
  P.ret.rec = P.d_ret; P.ret.nSlots = padded; P.ret.NB = best.NB; P.ret.NVs = best.NVs; P.ret.kMax = best.kMax;
  P.ret.rowLen = rowLen; P.ret.nPen = (best.kMax + 1) * rowLen;
  P.retGv = best.gv;
  P.retLdsBytes = ((best.gv ? 0 : (size_t)best.NB * best.NVs) + 2 * (size_t)(P.ret.nPen + nImp)) * sizeof(double) + WIDE_RET_TOKWIN * sizeof(int);
  P.retPeriod = best.period; P.retTauMax = best.tauMax; P.retPeriodMin = pMin;
  P.slotsPerColumn = padded; P.candsPerColumn = T.candsPerColumn; P.nSync = T.nSync;
  P.rounds = T.rounds;                                  // (planning tables: mb_machine_sweep_ops counts them)
  P.retOk = true;
  if (P.tbCodes) {      // (a part: the tables of its own states, local numbering; wide_parts_host joins them)
    // decode tables of the traceback codes: the candidate lists of the state nodes, in the order the planner laid them out
    // (kept on the host as well: mb_debug_wide_retimed hands them to the device-free simulation of tests/test_retimed_plan.py)
    std::vector<int> tbOff(SC + 1, 0);
    std::vector<std::vector<uint32_t>> ent(SC);
    bool fits = m->S < (1 << 15) && m->nTrans <= (1 << 16) && !P.backward && P.viterbi;
    for (const WNode &nd : best.nd) {
      const uint32_t x = nd.dst & WIDE_RET_NO_DST;
      if (x >= (uint32_t)SC) continue;                      // relays, constants, a part's imports and exports
      if (nd.t3.size() > 256) fits = false;
      for (const WCand &cd : nd.t3) {
        if (cd.ref == 0xFFFFFFFFu) { ent[x].push_back(0xFFFFFFFFu); continue; }      // the seed
        const uint32_t e = m->inPerm[cd.ref];
        ent[x].push_back((cd.ref << 16) | ((m->inTok[e] || m->outTok[e]) ? 1u << 15 : 0u) | m->src[e]);
      }
    }
    std::vector<uint32_t> flat;
    for (int x = 0; x < SC; ++x) { tbOff[x] = (int)flat.size(); flat.insert(flat.end(), ent[x].begin(), ent[x].end()); }
    tbOff[SC] = (int)flat.size();
    P.tbOk = fits && (hostOut || (up_w(P.d_tbOff, tbOff) && up_w(P.d_tbEntry, flat)));
    P.tbEntries = (long long)flat.size();
    P.h_tbOff = tbOff; P.h_tbEntry = flat;
  }
  if (verbose)
    fprintf(stderr, "[mbhip] wide retimed %s%s program: period %d (shortest %d), %d columns in flight, %zu rounds, %d slots per period (%d before padding; %lld candidates = %.0f %% of the lane slots), ring %d x %d (%d relays)%s, LDS %zu bytes\n",
            P.backward ? "backward" : "forward", P.viterbi ? " (max)" : "", best.period, pMin, best.kMax + 1, T.rounds.size(), padded, nSlots, T.candsPerColumn,
            100.0 * (double)T.candsPerColumn / (double)std::max<long long>(1, (long long)padded * W), best.NB, best.NVs, best.nRelay, best.gv ? " in L2" : "", P.retLdsBytes);
  return true;
}

bool wide_ret_host(const mb_machine *m, bool backward, bool viterbi, WideProgram &P, std::vector<WideRec> &stream, bool tbCodes) {
  P = WideProgram();
  P.backward = backward; P.viterbi = viterbi; P.tbCodes = tbCodes;
  P.W = env_int_w("MB_WIDE_LANES", m->S >= 192 ? 1024 : 256);
  std::vector<WNode> nodes;
  int nExtra = 0, nStages = 0; long long nPairs = 0;
  if (!wide_nodes(m, backward, 0, P.W, 1ll << 40, nodes, nExtra, nStages, nPairs)) return false;
  return wide_ret_build(m, P, nodes, (m->nOut ? m->nOut : m->nIn) + 1, &stream) && P.retOk;
}

// ---- k workgroups per sequence: the cut (see WidePartDev in mb_wide.h) -------------------------------------------------------------
namespace {
// strongly connected components of a graph in CSR form (Tarjan, explicit stack); returns their number, comp[v] = component of v
int scc_of(int n, const std::vector<int> &off, const std::vector<int> &adj, std::vector<int> &comp) {
  comp.assign(n, -1);
  std::vector<int> idx(n, -1), low(n, 0), stack, it(n, 0), call;
  std::vector<char> onStack(n, 0);
  int counter = 0, nComp = 0;
  for (int r = 0; r < n; ++r) {
    if (idx[r] >= 0) continue;
    call.push_back(r); idx[r] = low[r] = counter++; stack.push_back(r); onStack[r] = 1;
    while (!call.empty()) {
      const int v = call.back();
      if (it[v] < off[v + 1] - off[v]) {
        const int w = adj[off[v] + it[v]++];
        if (idx[w] < 0) { idx[w] = low[w] = counter++; stack.push_back(w); onStack[w] = 1; call.push_back(w); }
        else if (onStack[w]) low[v] = std::min(low[v], idx[w]);
      } else {
        call.pop_back();
        if (!call.empty()) low[call.back()] = std::min(low[call.back()], low[v]);
        if (low[v] == idx[v]) {
          for (;;) { const int w = stack.back(); stack.pop_back(); onStack[w] = 0; comp[w] = nComp; if (w == v) break; }
          ++nComp;
        }
      }
    }
  }
  return nComp;
}
}  // namespace

bool wide_parts_host(const mb_machine *m, bool backward, bool viterbi, bool tbCodes, int k, int W, std::vector<WidePartHost> &parts, int &nExpTot,
                     std::vector<int> *tbOffOut, std::vector<uint32_t> *tbEntryOut, WidePartHint *hint, int *WoutP, int *ringOutP) {
  parts.clear(); nExpTot = 0;
  int WoutL = 0, ringOutL = 8;
  int &Wout = WoutP ? *WoutP : WoutL, &ringOut = ringOutP ? *ringOutP : ringOutL;
  std::vector<std::vector<uint32_t>> tbOfState(tbCodes ? m->S : 0);
  const int S = m->S, nTok = (m->nOut ? m->nOut : m->nIn) + 1;
  if (k < 2 || S < 2 * k) return false;
  const bool verbose = opt_env("MB_WIDE_VERBOSE") != nullptr;
  std::vector<WNode> nodes;
  int nExtra = 0, nStages = 0; long long nPairs = 0;
  if (!wide_nodes(m, backward, 0, 1024, 1ll << 40, nodes, nExtra, nStages, nPairs)) return false;
  // the dependency graph of the sweep: source -> node, for every candidate the retimed program keeps (finite weight, a source that has
  // candidates itself)
  std::vector<char> live(S, 0);
  std::vector<int> nodeOf(S, -1);
  for (size_t i = 0; i < nodes.size(); ++i) { const int x = (int)(nodes[i].dst & W_IDX_MASK); live[x] = 1; nodeOf[x] = (int)i; }
  auto forCands = [&](const WNode &nd, const std::function<void(const WCand &)> &f) {
    for (const auto &l : nd.t2) for (const WCand &cd : l) f(cd);
    for (const WCand &cd : nd.t3) f(cd);
  };
  auto srcOf = [&](const WCand &cd) { return (int)(cd.src & 0x3fffffffu); };
  auto kept = [&](const WCand &cd) { const int y = srcOf(cd); return y < S && live[y] && cd.w != -INFINITY; };
  std::vector<int> off(S + 1, 0), adj, weight(S, 3);
  for (const WNode &nd : nodes) forCands(nd, [&](const WCand &cd) { if (kept(cd)) off[srcOf(cd) + 1]++; });
  for (int v = 0; v < S; ++v) off[v + 1] += off[v];
  adj.resize(off[S]);
  { std::vector<int> fillAt(off.begin(), off.end() - 1);
    for (const WNode &nd : nodes) { const int x = (int)(nd.dst & W_IDX_MASK); forCands(nd, [&](const WCand &cd) { if (kept(cd)) { adj[fillAt[srcOf(cd)]++] = x; weight[x]++; } }); } }
  std::vector<int> comp;
  const int nComp = scc_of(S, off, adj, comp);
  // a topological order of the components that follows the sweep's own order of states where it can (Kahn, the ready component with
  // the earliest state first), cut where the running weight (candidates + a round's share per state) passes i / k of the total
  std::vector<int> key(nComp, backward ? -1 : S), indeg(nComp, 0);
  std::vector<long long> cw(nComp, 0);
  for (int v = 0; v < S; ++v) { key[comp[v]] = backward ? std::max(key[comp[v]], v) : std::min(key[comp[v]], v); cw[comp[v]] += weight[v]; }
  std::vector<std::vector<int>> cadj(nComp);
  for (int v = 0; v < S; ++v)
    for (int a = off[v]; a < off[v + 1]; ++a) if (comp[adj[a]] != comp[v]) cadj[comp[v]].push_back(comp[adj[a]]);
  for (auto &l : cadj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); for (int c : l) indeg[c]++; }
  auto later = [&](int a, int b) { return backward ? key[a] < key[b] : key[a] > key[b]; };      // (priority_queue: the top is the one that is not "later")
  std::priority_queue<int, std::vector<int>, decltype(later)> ready(later);
  for (int c = 0; c < nComp; ++c) if (!indeg[c]) ready.push(c);
  long long total = 0, run = 0;
  for (long long w : cw) total += w;
  std::vector<int> partOfComp(nComp, 0);
  int cur = 0, seen = 0;
  while (!ready.empty()) {
    const int c = ready.top(); ready.pop(); ++seen;
    if (cur + 1 < k && run > 0 && run + cw[c] / 2 >= total * (cur + 1) / k) ++cur;
    partOfComp[c] = cur; run += cw[c];
    for (int d : cadj[c]) if (!--indeg[d]) ready.push(d);
  }
  if (seen != nComp) return false;
  const int K = cur + 1;
  if (K < 2) return false;
  std::vector<int> partOf(S), expIdx(S, -1), loc(S, -1);
  for (int v = 0; v < S; ++v) partOf[v] = partOfComp[comp[v]];
  std::vector<char> exported(S, 0);
  for (int v = 0; v < S; ++v)
    for (int a = off[v]; a < off[v + 1]; ++a) {
      if (partOf[adj[a]] < partOf[v]) return false;      // (cannot happen: the order is topological)
      if (partOf[adj[a]] > partOf[v]) exported[v] = 1;
    }
  std::vector<int> expIdx0(K + 1, 0);
  for (int p = 0; p < K; ++p) {
    expIdx0[p] = nExpTot;
    for (int v = 0; v < S; ++v) if (partOf[v] == p && exported[v]) expIdx[v] = nExpTot++;
  }
  expIdx0[K] = nExpTot;
  parts.resize(K);
  const int resultState = backward ? 0 : S - 1;
  // every part's local nodes (see RetPart), once
  struct Prep { std::vector<int> own, imps, exps; std::vector<WNode> ln; RetPart spec; };
  std::vector<Prep> prep(K);
  for (int p = 0; p < K; ++p) {
    Prep &Q = prep[p];
    std::vector<int> &own = Q.own, &imps = Q.imps, &exps = Q.exps;
    for (int v = 0; v < S; ++v) if (partOf[v] == p) { loc[v] = (int)own.size(); own.push_back(v); if (exported[v]) exps.push_back(v); }
    for (int v : own) if (nodeOf[v] >= 0) forCands(nodes[nodeOf[v]], [&](const WCand &cd) { if (kept(cd) && partOf[srcOf(cd)] != p) imps.push_back(srcOf(cd)); });
    std::sort(imps.begin(), imps.end()); imps.erase(std::unique(imps.begin(), imps.end()), imps.end());
    RetPart &spec = Q.spec; spec.nOwn = (int)own.size(); spec.nImp = (int)imps.size(); spec.nExp = (int)exps.size();
    const int SV = spec.nOwn + spec.nImp + spec.nExp;
    auto localOf = [&](const WCand &cd, WCand &out) -> bool {      // false: a candidate the program drops anyway (its source never holds a value)
      const int y = srcOf(cd);
      out = cd;
      if (y == S + 1) { out.src = (cd.src & 0xC0000000u) | (uint32_t)(SV + 1); return true; }
      if (y >= S || !live[y] || cd.w == -INFINITY) return false;
      const int v = partOf[y] == p ? loc[y] : spec.nOwn + (int)(std::lower_bound(imps.begin(), imps.end(), y) - imps.begin());
      out.src = (cd.src & 0xC0000000u) | (uint32_t)v;
      return true;
    };
    std::vector<WNode> &ln = Q.ln;
    for (int v : own) {
      if (nodeOf[v] < 0) continue;
      const WNode &nd = nodes[nodeOf[v]];
      WNode o{CUR(loc[v]), 0, {}, {}};
      o.t2.resize(nd.t2.size());
      WCand c2;
      for (size_t t = 0; t < nd.t2.size(); ++t) for (const WCand &cd : nd.t2[t]) if (localOf(cd, c2)) o.t2[t].push_back(c2);
      for (const WCand &cd : nd.t3) if (localOf(cd, c2)) o.t3.push_back(c2);
      ln.push_back(std::move(o));
    }
    for (int i = 0; i < spec.nImp; ++i) ln.push_back(WNode{CUR(spec.nOwn + i), 0, {}, {}});
    for (int j = 0; j < spec.nExp; ++j) ln.push_back(WNode{CUR(spec.nOwn + spec.nImp + j), 0, {}, {WCand{CUR(loc[exps[j]]), 0.0}}});
    if (hint && hint->valid && p < (int)hint->merged.size()) { spec.mergeKnown = true; spec.merged = hint->merged[p]; spec.nEdges = hint->nEdges[p]; }
    for (int v : own) loc[v] = -1;
  }
  // Lanes per part and ring depth: every part is planned for every candidate and the launch takes the pair whose SLOWEST part is fastest
  // under a model fitted on the 5 063-state machine cut in four (ms per 10 000 periods: 2.84 per round + 1.45 per slot of the padded
  // period, x 1 + 0.44 per eight wavefronts beyond eight; a ring of 4 saves padding and costs 2 %).  The two-transition candidates of
  // a part are chosen at the first candidate and kept.  A hint (the choice of an earlier build of the same cut) skips the search.
  // (a round of a sum program -- two butterflies, an exponential per lane, a logarithm -- is about twice a max program's: 110 against 60
  //  instructions; fitted on nothing but the Forward / Backward halves of the 5 063-state machine, 120 and 136 ms for 5 and 6 rounds)
  const double cRoundM = viterbi ? 2.84 : 5.7;
  struct Cand { int first, second; bool merge; };
  std::vector<Cand> cand;
  if (hint && hint->valid) cand.push_back({hint->W, hint->ring, hint->merge});
  else {
    const int ringEnv = env_int_w("MB_ONETAPE_PART_RING", 0);
    const bool mayMerge = env_int_w("MB_ONETAPE_PART_MERGE", 1) != 0;
    std::vector<int> Ws;
    if (W > 0) Ws.push_back(W);
    else for (int w : {384, 512, 640, 768, 896, 1024}) Ws.push_back(w);
    (void)ringEnv;      // (a ring of 4 was measured: within 3 % either way, not worth a second set of kernels)
    for (int mg = mayMerge ? 1 : 0; mg >= 0; --mg) for (int w : Ws) cand.push_back({w, 8, mg != 0});
  }
  const bool hinted = hint && hint->valid && (int)hint->period.size() == K;
  auto buildPart = [&](int p, int w, int r, bool mg, WideProgram &T, std::vector<WideRec> *stream) -> bool {
    Prep &Q = prep[p];
    Q.spec.ring = WIDE_RING; (void)r; Q.spec.useMerge = mg;
    T = WideProgram();
    T.backward = backward; T.viterbi = viterbi; T.tbCodes = tbCodes; T.W = w;
    std::vector<WideRec> scratch;
    // (a build from a kept choice -- a weight update -- also keeps every part's period: one relaxation, one plan)
    int keepPeriod = 0;
    if (hinted) { keepPeriod = hint->period[p]; T.retPeriodMin = hint->periodMin[p]; }
    return wide_ret_build(m, T, Q.ln, nTok, stream ? stream : &scratch, keepPeriod, &Q.spec) && T.retOk && !T.retGv && (!tbCodes || T.tbOk);
  };
  int bestW = 0, bestRing = 8; double bestCost = 1e300; bool bestMerge = true;
  if (cand.size() == 1) { bestW = cand[0].first; bestRing = cand[0].second; bestMerge = cand[0].merge; }
  else {
    const bool quiet = opt_env("MB_WIDE_VERBOSE_PARTS") == nullptr;      // (the search's own builds stay silent unless asked)
    const char *keepVerbose = opt_env("MB_WIDE_VERBOSE");
    if (quiet && keepVerbose) g_quiet_search = true;
    struct Res { bool ok = false; double cost = 0.0; int period = 0; };
    std::vector<Res> res(cand.size());
    for (size_t ci = 0; ci < cand.size(); ++ci) {
      const Cand &c = cand[ci];
      double worst = 0.0; bool ok = true; int maxPeriod = 0;
      for (int p = 0; p < K && ok; ++p) {
        WideProgram T;
        ok = buildPart(p, c.first, c.second, c.merge, T, nullptr);
        if (ok) {
          worst = std::max(worst, (cRoundM * (double)T.rounds.size() + 1.45 * (double)T.ret.nSlots) * (1.0 + 0.44 * std::max(0, c.first / 64 - 8) / 8.0) * (c.second == 4 ? 1.02 : 1.0));
          maxPeriod = std::max(maxPeriod, T.retPeriod);
        }
      }
      res[ci].ok = ok; res[ci].cost = worst; res[ci].period = maxPeriod;
    }
    for (size_t ci = 0; ci < cand.size(); ++ci) {
      const Cand &c = cand[ci];
      bool ok = res[ci].ok;
      // two-transition candidates lengthen the spans between a value and its readers: in a part whose ring nearly fills the LDS the relays
      // push the period back up (whole fn3 profile cut in four: period 13 where the one-transition program has 9) and the sweep is slower than
      // the model says -- taken only where they shorten the slowest part's period by a quarter at least
      if (ok && c.merge)
        for (size_t cj = 0; cj < cand.size(); ++cj)
          if (!cand[cj].merge && cand[cj].first == c.first && cand[cj].second == c.second && res[cj].ok && 4 * res[ci].period > 3 * res[cj].period) ok = false;
      if (verbose) fprintf(stderr, "[mbhip] wide parts: %d lanes, ring %d, %s-transition candidates: %s, slowest part's period %d, modelled %.1f\n", c.first, c.second, c.merge ? "two" : "one",
                           ok ? "taken into account" : (res[ci].ok ? "set aside" : "no program"), res[ci].period, res[ci].cost);
      if (ok && res[ci].cost < bestCost) { bestCost = res[ci].cost; bestW = c.first; bestRing = c.second; bestMerge = c.merge; }
    }
    if (quiet && keepVerbose) g_quiet_search = false;
    if (!bestW) { parts.clear(); return false; }
  }
  for (int p = 0; p < K; ++p) {
    Prep &Q = prep[p];
    const RetPart &spec = Q.spec;
    const std::vector<int> &own = Q.own, &imps = Q.imps;
    for (size_t q = 0; q < own.size(); ++q) loc[own[q]] = (int)q;
    WideProgram T;
    WidePartHost &H = parts[p];
    if (!buildPart(p, bestW, bestRing, bestMerge, T, &H.stream)) { parts.clear(); return false; }
    if (tbCodes) for (int q = 0; q < spec.nOwn; ++q) tbOfState[own[q]].assign(T.h_tbEntry.begin() + T.h_tbOff[q], T.h_tbEntry.begin() + T.h_tbOff[q + 1]);
    H.h = WidePartDev{};
    H.h.ret = T.ret; H.h.ret.rec = nullptr;
    H.h.w2Offset = (int)T.retW2Offset;
    H.h.Sloc = spec.nOwn; H.h.nImp = spec.nImp; H.h.expBase = spec.nOwn + spec.nImp + 2; H.h.expIdx0 = expIdx0[p]; H.h.nExp = spec.nExp;
    H.h.resultEntry = partOf[resultState] == p ? loc[resultState] : -1;
    H.tab.clear();
    for (int v : own) H.tab.push_back((uint32_t)v);
    for (int y : imps) H.tab.push_back((uint32_t)expIdx[y]);
    H.period = T.retPeriod; H.periodMin = T.retPeriodMin;
    H.modelCost = (cRoundM * (double)T.rounds.size() + 1.45 * (double)T.ret.nSlots) * (1.0 + 0.44 * std::max(0, bestW / 64 - 8) / 8.0);
    H.ldsBytes = ((T.retLdsBytes + 7) & ~(size_t)7) + 512 + 4 * (size_t)spec.nOwn;
    if (verbose)
      fprintf(stderr, "[mbhip] wide retimed part %d of %d: %d states, %d imports, %d exports, %d lanes, period %d, %zu rounds, %d slots per period, %d columns in flight, ring %d x %d, LDS %zu bytes\n",
              p, K, spec.nOwn, spec.nImp, spec.nExp, bestW, T.retPeriod, T.rounds.size(), T.ret.nSlots, T.ret.kMax + 1, T.ret.NB, T.ret.NVs, H.ldsBytes);
    for (int v : own) loc[v] = -1;
  }
  Wout = bestW; ringOut = bestMerge ? 1 : 0;      // (second output: the parts carry two-transition candidates)
  if (hint && !hint->valid) {
    hint->valid = true; hint->W = bestW; hint->ring = bestRing; hint->merge = bestMerge; hint->merged.clear(); hint->nEdges.clear();
    hint->period.clear(); hint->periodMin.clear();
    for (int p = 0; p < K; ++p) { hint->merged.push_back(prep[p].spec.merged); hint->nEdges.push_back(prep[p].spec.nEdges); hint->period.push_back(parts[p].period); hint->periodMin.push_back(parts[p].periodMin); }
  }
  if (tbCodes && tbOffOut && tbEntryOut) {
    tbOffOut->assign(S + 1, 0); tbEntryOut->clear();
    for (int v = 0; v < S; ++v) { (*tbOffOut)[v] = (int)tbEntryOut->size(); tbEntryOut->insert(tbEntryOut->end(), tbOfState[v].begin(), tbOfState[v].end()); }
    (*tbOffOut)[S] = (int)tbEntryOut->size();
  }
  return true;
}

bool wide_build(const mb_machine *m, bool backward, bool viterbi, WideProgram &P) {
  const bool haveShape = P.ok && P.shapeChosen;    // a weight refresh keeps the shape that was chosen
  const int keepStages = P.stages, keepPeriod = (P.ok && P.retOk) ? P.retPeriod : 0, keepPeriodMin = P.retPeriodMin;
  const bool keepTb = P.tbCodes;
  std::vector<WidePartHint> keepHints = std::move(P.partHints);      // (the cuts' choices depend on the graph: a weight update re-plans with them)
  wide_free(P);
  P.backward = backward; P.viterbi = viterbi; P.tbCodes = keepTb;
  P.partHints = std::move(keepHints);
  P.W = env_int_w("MB_WIDE_LANES", m->S >= 192 ? 1024 : 256);      // 509 states: 48 G cells/s with 1024 lanes, 35 with 256
  const int S = m->S, nLev = backward ? m->nLevB : m->nLevF;
  long long nSilent = 0;
  for (long long e = 0; e < m->nTrans; ++e) nSilent += (m->inTok[e] == 0 && m->outTok[e] == 0);
  const long long pairCap = std::max<long long>(64 * (nSilent + S), 1 << 16);
  const int want32 = env_int_w("MB_WIDE_FP32", -1);
  // The retimed sweep first (from the LEVELLED nodes): a machine that has one needs no column-by-column program, and a
  // weight refresh -- every EM iteration -- then costs one relaxation, one plan and one upload (20-node machine, two programs: 59 -> 5.6 ms on top of a 21 ms E-step; the first build 544 -> 51 ms).
  if (env_int_w("MB_WIDE_RETIMED", 1) && want32 <= 0) {      // (MB_WIDE_FP32 = 1 asks for the fp32 kernel)
    std::vector<WNode> levelled;
    int xe = 0, xs = 0; long long xp = 0;
    if (wide_nodes(m, backward, 0, P.W, pairCap, levelled, xe, xs, xp)) {
      P.retPeriodMin = keepPeriodMin;
      if (!wide_ret_build(m, P, levelled, (m->nOut ? m->nOut : m->nIn) + 1, nullptr, keepPeriod)) return false;
      if (P.retOk) {
        P.NV = S + 2; P.NX = 1; P.stages = 0;
        P.dev.S = S; P.dev.NV = P.NV; P.dev.NX = P.NX; P.dev.W = P.W;
        P.dev.resultIdx = backward ? 0 : S - 1; P.dev.backward = backward ? 1 : 0; P.dev.inputTape = m->nOut ? 0 : 1;
        P.ok = true; P.dirty = false;
        return true;
      }
    }
  }
  std::vector<WNode> nodes, bestNodes;
  int nExtra = 0, nStages = 0, bestExtra = 0, bestStages = 0, bestK = 0;
  long long nPairs = 0, bestPairs = 0;
  double best = 1e300;
  const bool verbose = opt_env("MB_WIDE_VERBOSE") != nullptr;
  auto consider = [&](int K) {
    if (!wide_nodes(m, backward, K, P.W, pairCap, nodes, nExtra, nStages, nPairs)) return false;
    const double c = wide_plan(nodes, nStages, (m->nOut ? m->nOut : m->nIn) + 1, P.W, false, nullptr);
    if (verbose) fprintf(stderr, "[mbhip] wide %s program, closure stages %d: modelled %.0f cycles per column (%lld pairs)\n", backward ? "backward" : "forward", K, c, nPairs);
    if (c < best) { best = c; bestNodes.swap(nodes); bestExtra = nExtra; bestStages = nStages; bestK = K; bestPairs = nPairs; }
    return true;
  };
  if (viterbi) consider(0);
  else {
    // MB_WIDE_CLOSURE_STAGES: K >= 0 uniform level groups (0 = levelled); -n = adaptive stages of at most n slots of candidates
    const int envK = env_int_w("MB_WIDE_CLOSURE_STAGES", -1000000);
    if (haveShape) consider(keepStages);
    else if (envK > -1000000) consider(envK >= 0 ? envK : (envK == -999 ? -(int)WIDE_STAGES_DP : envK * P.W));
    else {
      consider(0);
      for (int K = std::max(1, nLev - 1); K >= 1; K = (K * 2) / 3) {
        if (!consider(K)) break;
        if (K == 1) break;
      }
      if (env_int_w("MB_WIDE_ADAPTIVE_STAGES", 1)) consider(-(int)WIDE_STAGES_DP);
      if (env_int_w("MB_WIDE_ADAPTIVE_STAGES", 1))
        for (int q = 2; q <= 32; q += (q < 12 ? 1 : (q < 20 ? 2 : 4))) consider(-(q * P.W) / 4);      // 0.5 ... 8 slots of candidates per stage
    }
  }
  if (best >= 1e300) { set_error("wide program: no feasible shape"); return false; }
  P.stages = bestK; P.nPairs = bestPairs;
  P.NV = S + 2; P.NX = bestExtra + 1;
  P.dev.S = S;
  wide_plan(bestNodes, bestStages, (m->nOut ? m->nOut : m->nIn) + 1, P.W, true, &P);
  if (P.rounds.empty()) { set_error("wide program: empty machine"); return false; }
  // Runs of thin levels (a profile's delete chain: 2-20 states per level, hundreds of levels): when a stage and the next
  // one each fit the first wavefront, the LDS pipeline already orders that wavefront's store before its next read, so
  // the workgroup barrier between them is dropped; the other wavefronts run ahead to the barrier that ends the run.
  // (LDS columns only: the same-wavefront ordering is not relied upon for the L2 scratch vectors.)
  if (env_int_w("MB_WIDE_WAVE_LOCAL", 1) && P.vecBytes() <= WIDE_LDS_MAX && !env_int_w("MB_WIDE_GLOBAL_VECTORS", 0) && bestK == 0) {
    int dropped = 0;
    for (size_t r = 0; r + 1 < P.rounds.size(); ++r)
      if (P.rounds[r].sync && P.rounds[r].pad0 <= 64 && P.rounds[r + 1].pad0 <= 64 && P.rounds[r + 1].sync) { P.rounds[r].sync = 0; ++dropped; }
    P.nSync -= dropped;
  }
  static_assert(WIDE_RING == 8, "slot flags are packed eight to a 64-bit word");
  size_t nRecs = 0;
  // single precision relative to the column reference where it buys something: when two fp64 columns do not fit the LDS of
  // a CU (fp64 arithmetic is full rate on this chip: with both in LDS the fp64 kernel is the faster one, 39.9 vs 45.4 ms
  // on the 20-node profile machine).  MB_WIDE_FP32 = 1 / 0 forces it on / off.
  if (!viterbi && (want32 > 0 || (want32 < 0 && (size_t)(2 * P.NV + P.NX) * sizeof(double) > WIDE_LDS_MAX))) {
    std::vector<WideRec32> a32, b32;
    std::vector<unsigned long long> fw;
    // both columns in LDS when they fit; else the current one in LDS and the previous one in L2; else both in L2
    const size_t lds32 = WIDE_LDS_MAX - 64;
    const bool fits2 = (size_t)(2 * P.NV + P.NX) * sizeof(float) <= lds32;
    const int wantHyb = env_int_w("MB_WIDE_HYBRID", -1);
    P.hyb = (wantHyb > 0 || (wantHyb < 0 && !fits2)) && (size_t)(P.NV + P.NX) * sizeof(float) <= lds32 &&
            wide_linearise32(P, (m->nOut ? m->nOut : m->nIn) + 1, true, a32, b32, fw);
    if (P.hyb || wide_linearise32(P, (m->nOut ? m->nOut : m->nIn) + 1, false, a32, b32, fw)) {
      if (!up_w(P.d_seg32A, a32) || !up_w(P.d_seg32B, b32) || !up_w(P.d_flags, fw)) return false;
      P.f32 = true;
      nRecs = a32.size() + b32.size();
      P.dev32.segA = P.d_seg32A; P.dev32.segB = P.d_seg32B; P.dev32.flags = P.d_flags;
      P.dev32.S = S; P.dev32.NV = P.NV; P.dev32.NX = P.NX; P.dev32.W = P.W;
      P.dev32.resultIdx = backward ? 0 : S - 1; P.dev32.backward = backward ? 1 : 0; P.dev32.inputTape = m->nOut ? 0 : 1;
    }
  }
  if (!P.f32) {
    wide_linearise(P, (m->nOut ? m->nOut : m->nIn) + 1);
    if (!up_w(P.d_segA, P.segA) || !up_w(P.d_segB, P.segB)) return false;
    nRecs = P.segA.size() + P.segB.size();
    if (viterbi && !wide_vit_build(P, (m->nOut ? m->nOut : m->nIn) + 1)) return false;
  }
  P.dev.segA = P.d_segA; P.dev.segB = P.d_segB;
  P.dev.NV = P.NV; P.dev.NX = P.NX; P.dev.W = P.W;
  std::vector<WideRec>().swap(P.recs); std::vector<uint32_t>().swap(P.dsts);
  std::vector<WideRec>().swap(P.segA); std::vector<WideRec>().swap(P.segB);
  P.dev.resultIdx = backward ? 0 : S - 1;
  P.dev.backward = backward ? 1 : 0;
  P.dev.inputTape = m->nOut ? 0 : 1;
  P.ok = true; P.dirty = false; P.shapeChosen = true;
  if (verbose)
    fprintf(stderr, "[mbhip] wide %s%s program: %d stages, %zu rounds, %lld slots and %d barriers per column (%lld candidates = %.0f %% of the lane slots), %zu records, vectors %zu bytes\n",
            backward ? "backward" : "forward", viterbi ? " (max)" : "", bestK ? bestStages - 1 : 0, P.rounds.size(), P.slotsPerColumn, P.nSync, P.candsPerColumn,
            100.0 * (double)P.candsPerColumn / (double)std::max<long long>(1, P.slotsPerColumn * P.W), nRecs, P.vecBytes());
  if (verbose && viterbi) {   // shape of the levelled program: how much of a column is work of the first wavefront alone?
    long long thinRounds = 0, thinSlots = 0, wideRounds = 0, wideSlots = 0, tokRounds = 0; int maxLanes = 0;
    for (const WideRound &R : P.rounds) {
      if (R.tokStride) ++tokRounds;
      if (R.pad0 <= 64) { ++thinRounds; thinSlots += R.depth; } else { ++wideRounds; wideSlots += R.depth; }
      maxLanes = std::max(maxLanes, R.pad0);
    }
    fprintf(stderr, "[mbhip] wide (max) rounds: %lld fit the first wavefront (%lld slots), %lld do not (%lld slots, widest %d lanes), %lld read the previous column; %d phases\n",
            thinRounds, thinSlots, wideRounds, wideSlots, maxLanes, tokRounds, P.vit.nPhases);
  }
  return true;
}


template <int MODE, bool GV, bool FAST>
static int launch_wide(const WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_out, double *pool, double *loglike,
                       double *scratch, hipStream_t st, bool lastOnly, const WideProgram *P2 = nullptr, const PairDesc *d_desc2 = nullptr,
                       long long nPairs2 = 0, double *pool2 = nullptr, double *scratch2 = nullptr) {
  const size_t lds = GV ? 0 : std::max(P.vecBytes(), P2 ? P2->vecBytes() : (size_t)0);
  static bool attr = false;     // one flag per instantiation
  if (!GV && !attr) {
    MB_HIP(hipFuncSetAttribute((const void *)k_wide_sweep<MODE, GV, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
    attr = true;
  }
  WideDev dev = P.dev; dev.lastOnly = lastOnly ? 1 : 0;
  WideDev dev2 = P2 ? P2->dev : P.dev; dev2.lastOnly = dev.lastOnly;
  const WideSecond X{P2 ? (unsigned)nPairs : 0xFFFFFFFFu, d_desc2, pool2, scratch2};
  hipLaunchKernelGGL((k_wide_sweep<MODE, GV, FAST>), dim3((unsigned)(nPairs + nPairs2)), dim3(P.W), lds, st, dev, d_desc, d_out, pool, loglike, scratch, dev2, X);
  MB_HIP(hipGetLastError());
  return 0;
}

template <bool GV, bool HYB>
static int launch_wide32(const WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_out, double *pool, double *loglike,
                         float *scratch, hipStream_t st, bool lastOnly, const WideProgram *P2 = nullptr, const PairDesc *d_desc2 = nullptr,
                         long long nPairs2 = 0, double *pool2 = nullptr, float *scratch2 = nullptr) {
  auto ldsOf = [](const WideProgram &Q) { return HYB ? (size_t)(Q.NV + Q.NX) * sizeof(float) : Q.vecBytes32(); };
  const size_t lds = GV ? 0 : std::max(ldsOf(P), P2 ? ldsOf(*P2) : (size_t)0);
  static bool attr = false;
  if (!GV && !attr) {
    MB_HIP(hipFuncSetAttribute((const void *)k_wide_sum32<GV, HYB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX - 64));
    attr = true;
  }
  WideDev32 dev = P.dev32; dev.lastOnly = lastOnly ? 1 : 0;
  WideDev32 dev2 = P2 ? P2->dev32 : P.dev32; dev2.lastOnly = dev.lastOnly;
  const WideSecond X{P2 ? (unsigned)nPairs : 0xFFFFFFFFu, d_desc2, pool2, scratch2};
  hipLaunchKernelGGL((k_wide_sum32<GV, HYB>), dim3((unsigned)(nPairs + nPairs2)), dim3(P.W), lds, st, dev, d_desc, d_out, pool, loglike, scratch, dev2, X);
  MB_HIP(hipGetLastError());
  return 0;
}

static int g_last_parts = 1;
static bool g_last_jit = false;          // the last launch of this family ran the kernel generated for the machine (mb_wide_jit.cpp)
const char *wide_kernel_name(const WideProgram &P) {
  if (P.retOk && g_last_parts > 1) {      // (asked after the launch)
    static thread_local char nm[64];
    snprintf(nm, sizeof(nm), "%s<%s> in %d parts", g_last_jit ? "k_wide_jit" : "k_wide_retimed", P.viterbi ? (P.tbCodes ? "1,codes" : "1") : "0", g_last_parts);
    return nm;
  }
  if (P.retOk && g_last_jit) return P.viterbi ? (P.tbCodes ? "k_wide_jit<1,codes>" : "k_wide_jit<1>") : "k_wide_jit<0>";
  if (P.retOk) return P.retGv ? (P.viterbi ? (P.tbCodes ? "k_wide_retimed<1,L2,codes>" : "k_wide_retimed<1,L2>") : "k_wide_retimed<0,L2>")
                              : (P.viterbi ? (P.tbCodes ? "k_wide_retimed<1,codes>" : "k_wide_retimed<1>") : "k_wide_retimed<0>");
  if (P.f32) return "k_wide_sum32";
  if (P.viterbi) return P.vitOk ? "k_wide_viterbi" : "k_wide_sweep<1>";
  return "k_wide_sweep<0>";
}

// the next sum-semiring fills of retimed programs use the fp64 correction term (k_wide_retimed<.., ACC>): set by the E-step of long
// sequences (mb_api.hip, counts_chunks) around its two fills
static bool g_wide_accurate = false;
void wide_set_accurate(bool on) { g_wide_accurate = on; }

// ---- k workgroups per sequence: program sets and launches ------------------------------------------------------------------------
int wide_last_parts() { return g_last_parts; }
static bool g_parts_pending = false;        // a partitioned launch since the status was last read
static unsigned *g_part_err = nullptr;      // raised by a lane whose wait for an exchange value ran out (cleared on the stream in front of every partitioned launch)
// the programs with a partitioned launch in flight: when a wait runs out they are LATCHED to one workgroup per sequence (until they are
// rebuilt) and the API call is run once more -- a co-tenant on the device, a CU mask or two partitioned launches that do not fit the chip
// together cost one time-out, not every call that follows
static std::vector<WideProgram *> g_parts_inflight;
static bool g_parts_retry = false;

// k for a launch of nPairs sequences that may count on `cus` CUs: as many parts as fit, at most MB_ONETAPE_PARTS (0 or 1: off).
// Measured.  A machine whose ring fits the LDS of ONE CU (5 063 states): a part's period is bound by its chain of stages,
// not by its width -- with two-transition candidates 4 parts serve 64 sequences, 8 parts are 10 % faster than 4 for 8 sequences (Viterbi
// 25 vs 27.6 ms), 16 no better, two parts pay for the sum sweeps only: default at most 8.  A machine whose ring lives in L2 (21 761
// states): the parts bring it into LDS -- 16 sequences: Viterbi fill 160 -> 17 ms with 16 parts -- default at most 16.
static int wide_parts_k(const WideProgram &P, long long nPairs, int cus) {
  if (!P.retOk || nPairs <= 0 || cus <= 0 || P.partsOff) return 1;
  const int maxK = env_int_w("MB_ONETAPE_PARTS", P.retGv ? 16 : 8);
  const long long k = std::min<long long>(maxK, cus / nPairs);
  if (k < 2 || (k == 2 && P.viterbi && !P.retGv && !opt_env("MB_ONETAPE_PARTS"))) return 1;
  return (int)k;
}

static WidePartSet *wide_parts_get(const mb_machine *m, WideProgram &P, int k) {
  // lanes per part and ring depth: MB_ONETAPE_PART_LANES / MB_ONETAPE_PART_RING, else searched by wide_parts_host (0: its choice)
  int lanes = env_int_w("MB_ONETAPE_PART_LANES", 0);
  if (lanes < 64 || lanes > 1024 || lanes % 64) lanes = 0;
  const int ringAsk = env_int_w("MB_ONETAPE_PART_RING", 0) == 4 ? 4 : (env_int_w("MB_ONETAPE_PART_RING", 0) == 8 ? 8 : 0);
  for (WidePartSet &ps : P.partSets) if (ps.kWanted == k && ps.lanesAsked == lanes && ps.ringAsked == ringAsk) return ps.ok ? &ps : nullptr;
  P.partSets.emplace_back();
  WidePartSet &ps = P.partSets.back();
  ps.kWanted = k; ps.lanesAsked = lanes; ps.ringAsked = ringAsk;
  WidePartHint *hint = nullptr;      // what an earlier build of this cut chose (kept across weight updates: it depends on the graph)
  for (WidePartHint &h : P.partHints) if (h.kWanted == k && h.lanesAsked == lanes && h.ringAsked == ringAsk) hint = &h;
  if (!hint) { P.partHints.emplace_back(); hint = &P.partHints.back(); hint->kWanted = k; hint->lanesAsked = lanes; hint->ringAsked = ringAsk; }
  std::vector<WidePartHost> hp;
  std::vector<int> tbOff; std::vector<uint32_t> tbEntry;
  int mergeFlag = 0;
  if (!wide_parts_host(m, P.backward, P.viterbi, P.tbCodes, k, lanes, hp, ps.nExpTot, &tbOff, &tbEntry, hint, &ps.W, &mergeFlag)) { hint->valid = false; return nullptr; }
  ps.merge = mergeFlag != 0;
  if (P.tbCodes) {
    if (!up_w(ps.d_tbOff, tbOff) || !up_w(ps.d_tbEntry, tbEntry)) return nullptr;
    ps.tbEntries = (long long)tbEntry.size();
  }
  ps.k = (int)hp.size();
  ps.d_rec.assign(ps.k, nullptr); ps.d_tab.assign(ps.k, nullptr);
  ps.h_parts.resize(ps.k);
  for (int p = 0; p < ps.k; ++p) {
    if (!up_w(ps.d_rec[p], hp[p].stream) || !up_w(ps.d_tab[p], hp[p].tab)) return nullptr;
    WidePartDev d = hp[p].h;
    d.ret.rec = ps.d_rec[p]; d.gmap = ps.d_tab[p]; d.impIdx = ps.d_tab[p] + d.Sloc;
    ps.h_parts[p] = d;
    ps.ldsBytes = std::max(ps.ldsBytes, hp[p].ldsBytes);
    ps.period.push_back(hp[p].period); ps.slots.push_back(d.ret.nSlots);
    ps.modelCost = std::max(ps.modelCost, hp[p].modelCost);
  }
  for (int p = 0; p < ps.k; ++p) { ps.h_stream.push_back(std::move(hp[p].stream)); ps.h_tab.push_back(std::move(hp[p].tab)); }      // (the generated kernel's tables are built from them on first use)
  if (!up_w(ps.d_parts, ps.h_parts)) return nullptr;
  if (!g_part_err) {
    if (!hip_ok(hipMalloc((void **)&g_part_err, 256), "hipMalloc(part status)") || !hip_ok(hipMemset(g_part_err, 0, 256), "hipMemset(part status)")) { g_part_err = nullptr; return nullptr; }
  }
  ps.ok = true;
  return &ps;
}

// after the stream(s) of partitioned launches have been synchronised: did a wait run out?  (the flag is cleared)
bool wide_parts_failed() {
  if (!g_part_err || !g_parts_pending) return false;
  g_parts_pending = false;
  std::vector<WideProgram *> inflight;
  inflight.swap(g_parts_inflight);
  unsigned e = 0;
  if (hipMemcpy(&e, g_part_err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) { set_error("one-tape parts: status unreadable"); return true; }
  if (!e) return false;
  (void)hipMemset(g_part_err, 0, sizeof(e));
  for (WideProgram *P : inflight) P->partsOff = true;
  g_parts_retry = true;
  fprintf(stderr, "[mbhip] WARNING: a one-tape sweep with k workgroups per sequence waited longer than MB_ONETAPE_PART_TIMEOUT_S for a value of another part "
                  "(is the device shared, or a CU mask set?): results discarded, the machine's sweeps fall back to one workgroup per sequence\n");
  set_error("one-tape sweep with k workgroups per sequence: a workgroup waited longer than MB_ONETAPE_PART_TIMEOUT_S for a value of another part (results discarded)");
  return true;
}
// a call that failed because a wait ran out may be run once more: its programs are latched to one workgroup per sequence
bool wide_parts_retry() { const bool r = g_parts_retry; g_parts_retry = false; return r; }
// an API call that ends in an error must not leave a raised status word or a pending flag to the next call
void wide_parts_reset() {
  if (!g_parts_pending) return;
  g_parts_pending = false; g_parts_inflight.clear();
  if (g_part_err) (void)hipMemset(g_part_err, 0, sizeof(unsigned));
}

// mode: 0 fp64 cells / log-likelihood only, 1 traceback codes
static int wide_fill_parts(const mb_machine *m, WideProgram &P, WidePartSet &ps, const PairDesc *d_desc, const PairDesc *h_desc, long long nPairs, const int *d_tape,
                           double *pool, double *loglike, hipStream_t st, bool lastOnly, bool tb) {
  // exchange rows: one per column of every sequence; the buffer starts with the sequences' first rows (filled in on the device)
  long long rows = 0;
  for (long long p = 0; p < nPairs; ++p) rows += (long long)(m->nOut ? h_desc[p].outLen : h_desc[p].inLen) + 1;
  const size_t headBytes = ((size_t)nPairs * sizeof(long long) + 255) & ~(size_t)255, xBytes = (size_t)rows * (size_t)ps.nExpTot * sizeof(double);
  char *buf = (char *)ws_get(P.backward ? 14 : 13, headBytes + std::max<size_t>(xBytes, 8));
  if (!buf) return 1;
  hipLaunchKernelGGL(k_wide_part_rows, dim3(1), dim3(1), 0, st, d_desc, (int)nPairs, m->nOut ? 0 : 1, (long long *)buf);
  MB_HIP(hipMemsetAsync(buf + headBytes, 0xFF, std::max<size_t>(xBytes, 8), st));
  if (!g_parts_pending) MB_HIP(hipMemsetAsync(g_part_err, 0, sizeof(unsigned), st));      // (a second launch of the same call -- the other sweep on the other stream -- shares the word)
  WidePartArgs A{};
  A.parts = ps.d_parts; A.nSeq = (int)nPairs; A.nExpTot = ps.nExpTot; A.X = (double *)(buf + headBytes); A.xOff = (const long long *)buf;
  { const char *ts = opt_env("MB_ONETAPE_PART_TIMEOUT_S");      // seconds (a fraction is taken: the tests provoke the time-out with microseconds)
    const double secs = ts && *ts ? atof(ts) : 20.0;
    A.err = g_part_err; A.timeoutTicks = std::max<long long>(1, (long long)(secs * 1e8)); }
  if (std::find(g_parts_inflight.begin(), g_parts_inflight.end(), &P) == g_parts_inflight.end()) g_parts_inflight.push_back(&P);
  WideDev dev = P.dev; dev.lastOnly = lastOnly ? 1 : 0; dev.W = ps.W;
  // one workgroup per CU: a part's LDS is padded beyond half a CU's (the parts of a sequence are meant to run side by side on CUs of their own)
  size_t lds = ps.ldsBytes;
  if (env_int_w("MB_ONETAPE_PART_EXCLUSIVE", 1)) lds = std::max<size_t>(lds, 82 * 1024);
  if (lds > WIDE_LDS_MAX) { set_error("one-tape parts: LDS"); return 1; }
  const dim3 grid((unsigned)(nPairs * ps.k)), block((unsigned)ps.W);
  // the sweep generated for this machine and this cut (mb_wide_jit.cpp); the interpreter below is the fallback
  if (wide_jit_enabled()) {
    const bool acc = !P.viterbi && g_wide_accurate;
    WideJitKernel &J = ps.jit[acc ? 1 : 0];
    if (!J.tried) {
      std::vector<WideJitIn> ins(ps.k);
      for (int p = 0; p < ps.k; ++p) {
        const WidePartDev &h = ps.h_parts[p];
        WideJitIn &in = ins[p];
        in.ret = h.ret; in.W = ps.W; in.stream = ps.h_stream[p].data();
        in.w2 = ps.merge ? (const double *)((const char *)ps.h_stream[p].data() + h.w2Offset) : nullptr;
        in.part = true; in.S = h.Sloc; in.Sg = m->S; in.nImp = h.nImp; in.expBase = h.expBase; in.nExp = h.nExp; in.expIdx0 = h.expIdx0; in.resultEntry = h.resultEntry;
        in.gmap = ps.h_tab[p].data();
      }
      WideJitFlags F; F.viterbi = P.viterbi; F.tb = tb; F.acc = acc; F.backward = P.backward; F.inputTape = m->nOut == 0; F.nExpTot = ps.nExpTot;
      if (!wide_jit_build(ins, F, J, &J.why) && opt_env("MB_WIDE_VERBOSE")) fprintf(stderr, "[mbhip] wide jit (%d parts): interpreter kept -- %s\n", ps.k, J.why.c_str());
    }
    if (J.mod) {
      for (int p = 0; p < ps.k; ++p) J.args.impIdx[p] = ps.h_parts[p].impIdx;
      J.args.nSeq = A.nSeq; J.args.nExpTot = A.nExpTot; J.args.X = A.X; J.args.xOff = A.xOff; J.args.err = A.err; J.args.timeoutTicks = A.timeoutTicks;
      if (wide_jit_launch(J, dev, grid.x, lds, d_desc, d_tape, pool, loglike, st)) return 1;
      g_last_launches += 1;
      g_last_parts = ps.k;
      g_parts_pending = true;
      g_last_jit = true;
      return 0;
    }
  }
  g_last_jit = false;
#define WIDE_PART_GO(M, T, AC, W2F) do { \
    static bool attr = false; \
    if (!attr) { MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed_parts<M, T, AC, W2F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX)); attr = true; } \
    hipLaunchKernelGGL((k_wide_retimed_parts<M, T, AC, W2F>), grid, block, lds, st, dev, A, d_desc, d_tape, pool, loglike); } while (0)
#define WIDE_PART_W2(M, T, AC) do { if (ps.merge) WIDE_PART_GO(M, T, AC, true); else WIDE_PART_GO(M, T, AC, false); } while (0)
  if (P.viterbi) { if (tb) WIDE_PART_W2(MB_VITERBI, true, false); else WIDE_PART_W2(MB_VITERBI, false, false); }
  else if (g_wide_accurate) WIDE_PART_W2(MB_FORWARD, false, true);
  else WIDE_PART_W2(MB_FORWARD, false, false);
#undef WIDE_PART_W2
#undef WIDE_PART_GO
  MB_HIP(hipGetLastError());
  g_last_launches += 1;
  g_last_parts = ps.k;
  g_parts_pending = true;
  return 0;
}

// the exchange buffer of a launch (8 bytes x exports per column) must leave the device room for the matrices: at most a quarter of its memory
static bool wide_parts_fit(const mb_machine *m, const WidePartSet &ps, const PairDesc *h_desc, long long nPairs) {
  long long rows = 0;
  for (long long p = 0; p < nPairs; ++p) rows += (long long)(m->nOut ? h_desc[p].outLen : h_desc[p].inLen) + 1;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return false; }
  return (double)rows * (double)ps.nExpTot * 8.0 <= 0.25 * (double)totalB;
}

// Cutting a machine and planning its parts costs 0.5-1 s the first time (20-40 ms after a weight update): worth it for sweeps of
// thousands of columns -- from MB_ONETAPE_PARTS_MIN_LEN (4 096; 64 for a machine whose one-workgroup ring lives in L2, where the
// parts are 5-17 x faster) symbols in the longest sequence of the launch
static bool wide_parts_worth(const mb_machine *m, const WideProgram &P, const PairDesc *h_desc, long long nPairs) {
  const int minLen = env_int_w("MB_ONETAPE_PARTS_MIN_LEN", P.retGv ? 64 : 4096);
  for (long long p = 0; p < nPairs; ++p) if ((m->nOut ? h_desc[p].outLen : h_desc[p].inLen) >= minLen) return true;
  return false;
}

// modelled time per column of the cut such a launch would use (its slowest part; 0: no cut) -- what a sequence cut in two is divided by
double wide_parts_cost(const mb_machine *m, WideProgram &P, long long nPairs, int cus, const PairDesc *h_desc) {
  if (h_desc && !wide_parts_worth(m, P, h_desc, nPairs)) return 0.0;
  const int k = wide_parts_k(P, nPairs, cus);
  if (k < 2) return 0.0;
  WidePartSet *ps = wide_parts_get(m, P, k);
  return ps ? ps->modelCost : 0.0;
}

int wide_parts_for(const mb_machine *m, WideProgram &P, long long nPairs, int cus, const PairDesc *h_desc) {
  if (h_desc && !wide_parts_worth(m, P, h_desc, nPairs)) return 1;
  const int k = wide_parts_k(P, nPairs, cus);
  if (k < 2) return 1;
  WidePartSet *ps = wide_parts_get(m, P, k);
  if (ps && h_desc && !wide_parts_fit(m, *ps, h_desc, nPairs)) return 1;
  return ps ? ps->k : 1;
}

// the generated kernel of the ONE-workgroup retimed program (ring in LDS), built on first use; nullptr: the interpreter keeps the program
static WideJitKernel *wide_jit_one(const mb_machine *m, WideProgram &P, bool tb, bool acc) {
  if (!P.retOk || P.retGv || P.h_ret.empty() || !wide_jit_enabled()) return nullptr;
  WideJitKernel &J = P.jit[acc ? 1 : 0];
  if (!J.tried) {
    std::vector<WideJitIn> ins(1);
    WideJitIn &in = ins[0];
    in.ret = P.ret; in.W = P.W; in.stream = P.h_ret.data(); in.S = m->S; in.Sg = m->S; in.resultEntry = P.dev.resultIdx;
    WideJitFlags F; F.viterbi = P.viterbi; F.tb = tb; F.acc = acc; F.backward = P.backward; F.inputTape = m->nOut == 0;
    if (!wide_jit_build(ins, F, J, &J.why) && opt_env("MB_WIDE_VERBOSE")) fprintf(stderr, "[mbhip] wide jit (one workgroup per sequence): interpreter kept -- %s\n", J.why.c_str());
  }
  return J.mod ? &J : nullptr;
}

int wide_fill(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_tape, double *pool,
              double *loglike, hipStream_t st, bool lastOnly, const PairDesc *h_desc, int cus) {
  const int *d_out = d_tape;       // the token array of the machine's one tape (outputs of a generator, inputs of a recogniser)
  if (!P.ok) { set_error("wide program not built"); return 1; }
  if (nPairs <= 0) return 0;
  // L2-resident column vectors of the kernels whose columns do not fit the LDS: from the library's workspaces (one slot per
  // sweep direction, so that a Forward and a Backward sweep may run side by side on two streams); nothing here waits for
  // the device
  const int scratchSlot = P.backward ? 12 : 11;
  g_last_parts = 1; g_last_jit = false;
  if (P.retOk && h_desc && wide_parts_worth(m, P, h_desc, nPairs)) {
    const int k = wide_parts_k(P, nPairs, cus);
    WidePartSet *ps = k >= 2 ? wide_parts_get(m, P, k) : nullptr;
    if (ps && wide_parts_fit(m, *ps, h_desc, nPairs)) return wide_fill_parts(m, P, *ps, d_desc, h_desc, nPairs, d_tape, pool, loglike, st, lastOnly, false);
  }
  if (P.retOk && !P.retGv) {
    const bool acc = g_wide_accurate && !P.viterbi;
    if (WideJitKernel *J = wide_jit_one(m, P, false, acc)) {
      WideDev dev = P.dev; dev.lastOnly = lastOnly ? 1 : 0;
      if (wide_jit_launch(*J, dev, (unsigned)nPairs, 0, d_desc, d_out, pool, loglike, st)) return 1;
      g_last_launches += 1;
      g_last_jit = true;
      return 0;
    }
  }
  if (P.retOk) {
    static bool attr = false;
    if (!attr) {
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_VITERBI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_FORWARD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_VITERBI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_FORWARD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_FORWARD, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_FORWARD, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
      attr = true;
    }
    double *ringScratch = nullptr;
    if (P.retGv && !(ringScratch = (double *)ws_get(scratchSlot, (size_t)nPairs * (size_t)P.ret.NB * P.ret.NVs * sizeof(double)))) return 1;
    WideDev dev = P.dev; dev.lastOnly = lastOnly ? 1 : 0;
    const size_t accLds = ((P.retLdsBytes + 7) & ~(size_t)7) + 512;      // + the table of 2^(j/64)
    if (g_wide_accurate && !P.viterbi && accLds <= WIDE_LDS_MAX) {
      if (P.retGv) hipLaunchKernelGGL((k_wide_retimed<MB_FORWARD, true, false, true>), dim3((unsigned)nPairs), dim3(P.W), accLds, st, dev, P.ret, d_desc, d_out, pool, loglike, ringScratch);
      else hipLaunchKernelGGL((k_wide_retimed<MB_FORWARD, false, false, true>), dim3((unsigned)nPairs), dim3(P.W), accLds, st, dev, P.ret, d_desc, d_out, pool, loglike, ringScratch);
      MB_HIP(hipGetLastError());
      g_last_launches += 1;
      return 0;
    }
#define WIDE_RET_GO(M, G) hipLaunchKernelGGL((k_wide_retimed<M, G>), dim3((unsigned)nPairs), dim3(P.W), P.retLdsBytes, st, dev, P.ret, d_desc, d_out, pool, loglike, ringScratch)
    if (P.viterbi) { if (P.retGv) WIDE_RET_GO(MB_VITERBI, true); else WIDE_RET_GO(MB_VITERBI, false); }
    else { if (P.retGv) WIDE_RET_GO(MB_FORWARD, true); else WIDE_RET_GO(MB_FORWARD, false); }
#undef WIDE_RET_GO
    MB_HIP(hipGetLastError());
    g_last_launches += 1;
    return 0;
  }
  if (P.f32) {
    const bool gv32 = !P.hyb && (P.vecBytes32() > WIDE_LDS_MAX - 64 || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0));
    float *scr = nullptr;
    if (gv32 && !(scr = (float *)ws_get(scratchSlot, (size_t)nPairs * P.vecBytes32()))) return 1;
    if (P.hyb && !(scr = (float *)ws_get(scratchSlot, (size_t)nPairs * P.NV * sizeof(float)))) return 1;
    const int rc32 = P.hyb ? launch_wide32<false, true>(P, d_desc, nPairs, d_out, pool, loglike, scr, st, lastOnly)
                           : (gv32 ? launch_wide32<true, false>(P, d_desc, nPairs, d_out, pool, loglike, scr, st, lastOnly)
                                   : launch_wide32<false, false>(P, d_desc, nPairs, d_out, pool, loglike, scr, st, lastOnly));
    g_last_launches += 1;
    return rc32;
  }
  if (P.viterbi && P.vitOk) {
    static bool attr = false;
    if (!attr) { MB_HIP(hipFuncSetAttribute((const void *)k_wide_viterbi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX)); attr = true; }
    WideDev dev = P.dev; dev.lastOnly = lastOnly ? 1 : 0;
    hipLaunchKernelGGL(k_wide_viterbi, dim3((unsigned)nPairs), dim3(P.W), P.vecBytes(), st, dev, P.vit, d_desc, d_out, pool, loglike);
    MB_HIP(hipGetLastError());
    g_last_launches += 1;
    return 0;
  }
  const bool gv = P.vecBytes() > WIDE_LDS_MAX || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0);
  double *scratch = nullptr;
  if (gv && !(scratch = (double *)ws_get(scratchSlot, (size_t)nPairs * P.vecBytes()))) return 1;
  int rc;
#define WIDE_GO(M, G, F) launch_wide<M, G, F>(P, d_desc, nPairs, d_out, pool, loglike, scratch, st, lastOnly)
  if (P.viterbi) rc = gv ? (P.fastIdx ? WIDE_GO(MB_VITERBI, true, true) : WIDE_GO(MB_VITERBI, true, false))
                         : (P.fastIdx ? WIDE_GO(MB_VITERBI, false, true) : WIDE_GO(MB_VITERBI, false, false));
  else rc = gv ? (P.fastIdx ? WIDE_GO(MB_FORWARD, true, true) : WIDE_GO(MB_FORWARD, true, false))
               : (P.fastIdx ? WIDE_GO(MB_FORWARD, false, true) : WIDE_GO(MB_FORWARD, false, false));
#undef WIDE_GO
  g_last_launches += 1;
  return rc;
}

int wide_fill_tb(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_tape, unsigned char *tb,
                 double *loglike, hipStream_t st, const PairDesc *h_desc, int cus) {
  if (!P.ok || !P.retOk || !P.tbOk || !P.viterbi || P.backward) { set_error("one-tape traceback-code program not built"); return 1; }
  if (nPairs <= 0) return 0;
  g_last_parts = 1; g_last_jit = false;
  if (h_desc && wide_parts_worth(m, P, h_desc, nPairs)) {
    const int k = wide_parts_k(P, nPairs, cus);
    WidePartSet *ps = k >= 2 ? wide_parts_get(m, P, k) : nullptr;
    if (ps && wide_parts_fit(m, *ps, h_desc, nPairs)) { P.tbFromSet = (int)(ps - P.partSets.data()); return wide_fill_parts(m, P, *ps, d_desc, h_desc, nPairs, d_tape, (double *)tb, loglike, st, false, true); }
  }
  P.tbFromSet = -1;
  if (WideJitKernel *J = wide_jit_one(m, P, true, false)) {
    WideDev dev = P.dev; dev.lastOnly = 0;
    if (wide_jit_launch(*J, dev, (unsigned)nPairs, 0, d_desc, d_tape, (double *)tb, loglike, st)) return 1;
    g_last_launches += 1;
    g_last_jit = true;
    return 0;
  }
  static bool attr = false;
  if (!attr) {
    MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_VITERBI, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
    MB_HIP(hipFuncSetAttribute((const void *)k_wide_retimed<MB_VITERBI, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
    attr = true;
  }
  double *ringScratch = nullptr;
  if (P.retGv && !(ringScratch = (double *)ws_get(11, (size_t)nPairs * (size_t)P.ret.NB * P.ret.NVs * sizeof(double)))) return 1;
  WideDev dev = P.dev; dev.lastOnly = 0;
  if (P.retGv) hipLaunchKernelGGL((k_wide_retimed<MB_VITERBI, true, true>), dim3((unsigned)nPairs), dim3(P.W), P.retLdsBytes, st, dev, P.ret, d_desc, d_tape, (double *)tb, loglike, ringScratch);
  else hipLaunchKernelGGL((k_wide_retimed<MB_VITERBI, false, true>), dim3((unsigned)nPairs), dim3(P.W), P.retLdsBytes, st, dev, P.ret, d_desc, d_tape, (double *)tb, loglike, ringScratch);
  MB_HIP(hipGetLastError());
  g_last_launches += 1;
  return 0;
}

// ---- DPMatrix::traceBack over the codes (src/dpmatrix.defs.h:82-110) --------------------------------------------------------
// One workgroup per sequence.  A step reads the code of (column, state), decodes it to (transition, source state, emitting?) and
// moves -- a chain of dependent look-ups, so everything it touches sits in LDS: a WINDOW of code rows (the path goes back at
// most one column per step) in two halves, one walked by the first lane while the other wavefronts fetch the rows below it,
// and the decode tables when they fit beside the window (else they are read through L2).  The path is written backwards as
// positions of the incoming view; a second kernel turns them into edge ids (off the walker's chain).
struct WideTbWalk { const int *tbOff; const uint32_t *tbEntry; long long nEntries; int tablesInLds, rowsPerHalf, fast; };      // (in LDS the offsets are 16-bit: nEntries < 65536 there)

// `fast` (tables in LDS and 8 more bytes per state fit beside them): the low halves -- emitting << 15 | source state -- of a state's
// first FOUR entries sit in one 8-byte LDS word, read together with the code, so the chain from one step to the next is ONE LDS
// round trip (code and word in parallel, a shift) instead of two (code, then entry); the walk records the entry's INDEX, which
// k_onetape_path_ids turns into the edge id afterwards, in parallel, off the chain.  A code >= 4, the seed
// and a full slot take the plain step below.
__global__ __launch_bounds__(256) void k_onetape_traceback_codes(DevMachine m, WideTbWalk Q, const PairDesc *__restrict__ pairs, int inputTape,
                                                                const unsigned char *__restrict__ tb, const double *__restrict__ loglike,
                                                                const long long *__restrict__ slotOff, uint32_t *__restrict__ pathBuf,
                                                                long long *__restrict__ pathLen) {
  extern __shared__ unsigned int twl[];
  const long long p = blockIdx.x;
  const PairDesc pd = pairs[p];
  const int S = m.S, Sb = (S + 3) & ~3, L = inputTape ? pd.inLen : pd.outLen, tid = threadIdx.x;
  const int R = Q.rowsPerHalf, rowW = Sb >> 2;
  unsigned int *win = twl;                                         // [2][R][rowW]: half h holds columns hi(h) - R + 1 .. hi(h)
  const size_t winWords = (2 * (size_t)R * rowW + 1) & ~(size_t)1;
  unsigned long long *fastL = (unsigned long long *)(twl + winWords);   // (fast) [S]: 4 x 16 bits, 0xFFFF = no such entry / the seed
  unsigned int *entL = twl + winWords + (Q.fast ? 2 * (size_t)S : 0);   // decode tables (tablesInLds): entries, then 16-bit offsets
  unsigned short *offL = (unsigned short *)(entL + Q.nEntries);
  if (!(loglike[p] > -INFINITY)) { if (tid == 0) pathLen[p] = -1; return; }      // no path: src/dpmatrix.defs.h:84
  if (Q.tablesInLds) {
    for (int k = tid; k <= S; k += 256) offL[k] = (unsigned short)Q.tbOff[k];
    for (long long k = tid; k < Q.nEntries; k += 256) entL[k] = Q.tbEntry[k];
    if (Q.fast)
      for (int k = tid; k < S; k += 256) {
        const int o0 = Q.tbOff[k], cnt = Q.tbOff[k + 1] - o0;
        unsigned long long w = 0ull;
        for (int j = 0; j < 4; ++j) w |= (unsigned long long)(j < cnt ? (Q.tbEntry[o0 + j] & 0xFFFFu) : 0xFFFFu) << (16 * j);
        fastL[k] = w;
      }
  }
  const unsigned int *rows = (const unsigned int *)(tb + pd.cellBase);      // (cellBase and Sb are multiples of 4)
  auto fill = [&](int half, int hi, int first, int nThreads) {    // columns hi - R + 1 .. hi (those that exist) into half `half`, by threads first .. first + nThreads - 1
    const int lo = max(hi - R + 1, 0);
    if (hi < 0) return;
    const unsigned int *src = rows + (long long)lo * rowW;
    unsigned int *dst = win + (size_t)half * R * rowW + (size_t)(lo - (hi - R + 1)) * rowW;
    const int n = (hi - lo + 1) * rowW, nv = n >> 2;
    // 16 bytes per lane and load, eight loads in flight per lane: with single dwords, four at a time, a half of four 5 KB rows was
    // seven round trips to memory one after the other (3 us) -- THAT was the walker's time, not its walk (rows are 4-byte aligned:
    // global_load_dwordx4 takes that, the LDS side is written dword by dword)
    typedef unsigned int u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
    for (int k0 = tid - first; k0 < nv; k0 += nThreads * 8) {
      u32x4a4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + u * nThreads; if (k < nv) v[u] = ((const u32x4a4 *)src)[k]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + u * nThreads; if (k < nv) ((u32x4a4 *)dst)[k] = v[u]; }
    }
    for (int k = (nv << 2) + tid - first; k < n; k += nThreads) dst[k] = src[k];
  };
  // two words behind everything else: the walk's status after half w, in word w & 1 (every thread leaves the loop with it)
  int *flag = (int *)(Q.tablesInLds ? (char *)offL + ((((size_t)S + 1) * 2 + 3) & ~(size_t)3) : (char *)(twl + winWords));
  if (tid < 2) flag[tid] = 0;
  fill(0, L, 0, 256);
  __syncthreads();
  uint32_t *out = pathBuf + slotOff[p + 1];
  const long long cap = slotOff[p + 1] - slotOff[p];
  long long n = 0;
  int c = L, s = S - 1, status = 0;                               // status: 0 walking, 1 done, -2 slot full, -3 dead end
  // The walk belongs to the first wavefront, ALL of its lanes: its state is uniform, every look-up comes back through
  // readfirstlane, so position, state, counters and branches live on the scalar unit (a dependent scalar instruction issues every
  // cycle or two, a dependent vector one every four to eight, and a branch needs no exec mask) -- only the LDS addresses and the
  // value lane 0 stores pass through vector registers.  (As one lane's vector code a step was ~60 instructions and 8 masked branches.)
#define RFL(x) ((unsigned int)__builtin_amdgcn_readfirstlane((int)(x)))
  for (int w = 0, hi = L; hi >= 0; ++w, hi -= R) {
    const int half = w & 1;
    if (RFL(tid >> 6) != 0u) fill(half ^ 1, hi - R, 64, 192);      // the rows below this half, fetched while it is walked (a UNIFORM branch: the walk's state must not meet a per-thread join)
    else {
      const unsigned char *hb = (const unsigned char *)(win + (size_t)half * R * rowW);
      const int base = hi - R + 1;                                 // column of the half's first row
      while (status == 0 && c >= base) {
        if (c == 0 && s == 0) { status = 1; break; }
        const unsigned int codeV = hb[(c - base) * Sb + s];        // (every look-up of the step is requested before the first one is awaited)
        unsigned int code;
        if (Q.fast) {
          const unsigned long long fw = fastL[s];
          const unsigned int offV = offL[s];
          code = RFL(codeV);
          const unsigned int fwLo = RFL((unsigned int)fw), fwHi = RFL((unsigned int)(fw >> 32));
          const int o0f = (int)RFL(offV);
          const unsigned int f16 = code < 4u ? (((code & 2u) ? fwHi : fwLo) >> (16u * (code & 1u))) & 0xFFFFu : 0xFFFFu;
          if (f16 != 0xFFFFu && n < cap && !((f16 & 0x8000u) && c == 0)) {
            ++n;
            if (tid == 0) out[-n] = (uint32_t)(o0f + (int)code);    // the ENTRY's index: k_onetape_path_ids turns it into the edge id, off the walk
            s = (int)(f16 & 0x7fffu);
            c -= (int)(f16 >> 15);
            continue;
          }
        } else code = RFL(codeV);
        const int o0 = (int)RFL(Q.tablesInLds ? (int)offL[s] : Q.tbOff[s]), o1 = (int)RFL(Q.tablesInLds ? (int)offL[s + 1] : Q.tbOff[s + 1]);
        if ((int)code >= o1 - o0) { status = -3; break; }
        const unsigned int e = RFL(Q.tablesInLds ? entL[o0 + code] : Q.tbEntry[o0 + code]);
        if (e == 0xFFFFFFFFu) { status = (c == 0) ? 1 : -3; break; }      // the seed: cell (0, start)
        if (n >= cap) { status = -2; break; }
        ++n;
        if (tid == 0) out[-n] = (uint32_t)(o0 + (int)code);        // the entry's index (edge ids: k_onetape_path_ids)
        s = (int)(e & 0x7fffu);
        if (e & 0x8000u) { if (c == 0) { status = -3; break; } --c; }
      }
      if (tid == 0) flag[half] = status;
    }
    __syncthreads();
    if (flag[half] != 0) break;                                    // (word w & 1 is written again two halves on, behind the next barrier)
  }
#undef RFL
  if (tid == 0) pathLen[p] = status == 1 ? n : (status == 0 ? (c == 0 && s == 0 ? n : -3) : status);
}

// entry index -> position in the incoming view (the entry's high half) -> edge id
__global__ __launch_bounds__(256) void k_onetape_path_ids(DevMachine m, const uint32_t *__restrict__ tbEntry, const long long *__restrict__ slotOff,
                                                         const long long *__restrict__ pathLen, uint32_t *__restrict__ pathBuf) {
  const long long p = blockIdx.x, n = pathLen[p];
  if (n <= 0) return;
  uint32_t *q = pathBuf + slotOff[p + 1] - n;
  for (long long k = threadIdx.x; k < n; k += 256) q[k] = m.inEid[tbEntry[q[k]] >> 16];
}

int wide_traceback_codes(const mb_machine *m, const WideProgram &P, const PairDesc *d_pairs, long long nPairs, const unsigned char *tb,
                         const double *d_loglike, const long long *d_slotOff, uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st) {
  if (!P.tbOk) { set_error("one-tape traceback-code program not built"); return 1; }
  if (nPairs <= 0) return 0;
  const int S = m->S, Sb = wide_tb_stride(S);
  // (codes written by k workgroups per sequence decode with that cut's tables: two-transition candidates change the places)
  const bool fromSet = P.tbFromSet >= 0 && P.tbFromSet < (int)P.partSets.size();
  const int *tbOffD = fromSet ? P.partSets[P.tbFromSet].d_tbOff : P.d_tbOff;
  const uint32_t *tbEntryD = fromSet ? P.partSets[P.tbFromSet].d_tbEntry : P.d_tbEntry;
  const long long tbEntriesN = fromSet ? P.partSets[P.tbFromSet].tbEntries : P.tbEntries;
  size_t tabBytes = (((size_t)(S + 1) * 2 + 3) & ~(size_t)3) + (size_t)tbEntriesN * 4;
  WideTbWalk Q{tbOffD, tbEntryD, tbEntriesN, 0, 1, 0};
  const size_t budget = 150 * 1024;
  // the window wants at least 2 x 2 rows; the tables go to LDS when 2 x 4 rows still fit beside them, the fast words (8 bytes per
  // state) when they do too
  if (tabBytes + (size_t)8 * Sb <= budget && tbEntriesN < 65536) Q.tablesInLds = 1;
  if (Q.tablesInLds && tabBytes + (size_t)S * 8 + 8 + (size_t)8 * Sb <= budget && env_int_w("MB_ONETAPE_TB_FAST", 1) != 0) { Q.fast = 1; tabBytes += (size_t)S * 8 + 8; }
  const size_t forRows = budget - (Q.tablesInLds ? tabBytes : 0);
  Q.rowsPerHalf = (int)std::max<size_t>(1, std::min<size_t>(forRows / (2 * (size_t)Sb), 32));
  if ((size_t)2 * Q.rowsPerHalf * Sb > budget) { set_error("one-tape traceback codes: a code row exceeds the LDS"); return 1; }
  const size_t lds = (size_t)2 * Q.rowsPerHalf * Sb + 8 + (Q.tablesInLds ? tabBytes : 0) + 16;      // (+ two status words)
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)k_onetape_traceback_codes, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(k_onetape_traceback_codes, dim3((unsigned)nPairs), dim3(256), lds, st, m->dev, Q, d_pairs, m->nIn != 0 ? 1 : 0, tb, d_loglike, d_slotOff, d_pathBuf, d_pathLen);
  hipLaunchKernelGGL(k_onetape_path_ids, dim3((unsigned)nPairs), dim3(256), 0, st, m->dev, tbEntryD, d_slotOff, (const long long *)d_pathLen, d_pathBuf);
  return hip_ok(hipGetLastError(), "one-tape traceback (codes) launch") ? 0 : 1;
}

int wide_fill2(const mb_machine *m, WideProgram &A, WideProgram &B, const PairDesc *d_descA, const PairDesc *d_descB, long long nA, long long nB,
               const int *d_tape, double *poolA, double *poolB, hipStream_t st, bool lastOnly) {
  (void)m;
  if (!A.ok || !B.ok) { set_error("wide program not built"); return 1; }
  if (nA <= 0 || nB <= 0) return -1;
  if (A.viterbi || B.viterbi || A.retOk || B.retOk || A.f32 != B.f32 || A.hyb != B.hyb || A.fastIdx != B.fastIdx || A.W != B.W) return -1;
  if (A.f32) {
    auto gvOf = [](const WideProgram &Q) { return !Q.hyb && (Q.vecBytes32() > WIDE_LDS_MAX - 64 || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0)); };
    if (gvOf(A) != gvOf(B)) return -1;
    const bool gv32 = gvOf(A);
    float *sa = nullptr, *sb = nullptr;
    if (gv32 && (!(sa = (float *)ws_get(11, (size_t)nA * A.vecBytes32())) || !(sb = (float *)ws_get(12, (size_t)nB * B.vecBytes32())))) return 1;
    if (A.hyb && (!(sa = (float *)ws_get(11, (size_t)nA * A.NV * sizeof(float))) || !(sb = (float *)ws_get(12, (size_t)nB * B.NV * sizeof(float))))) return 1;
    const int rc = A.hyb ? launch_wide32<false, true>(A, d_descA, nA, d_tape, poolA, nullptr, sa, st, lastOnly, &B, d_descB, nB, poolB, sb)
                         : (gv32 ? launch_wide32<true, false>(A, d_descA, nA, d_tape, poolA, nullptr, sa, st, lastOnly, &B, d_descB, nB, poolB, sb)
                                 : launch_wide32<false, false>(A, d_descA, nA, d_tape, poolA, nullptr, sa, st, lastOnly, &B, d_descB, nB, poolB, sb));
    g_last_launches += 1;
    return rc;
  }
  auto gvOf = [](const WideProgram &Q) { return Q.vecBytes() > WIDE_LDS_MAX || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0) != 0; };
  if (gvOf(A) != gvOf(B)) return -1;
  const bool gv = gvOf(A);
  double *sa = nullptr, *sb = nullptr;
  if (gv && (!(sa = (double *)ws_get(11, (size_t)nA * A.vecBytes())) || !(sb = (double *)ws_get(12, (size_t)nB * B.vecBytes())))) return 1;
  int rc;
#define WIDE_GO2(G, F) launch_wide<MB_FORWARD, G, F>(A, d_descA, nA, d_tape, poolA, nullptr, sa, st, lastOnly, &B, d_descB, nB, poolB, sb)
  rc = gv ? (A.fastIdx ? WIDE_GO2(true, true) : WIDE_GO2(true, false)) : (A.fastIdx ? WIDE_GO2(false, true) : WIDE_GO2(false, false));
#undef WIDE_GO2
  g_last_launches += 1;
  return rc;
}

// ------------------------------------------------------------------------------------------------------------
// a sequence cut in two: Forward over the prefix and Backward over the suffix run side by side, joined here
// ------------------------------------------------------------------------------------------------------------
// P(sequence) = sum over the EMITTING transitions (s -> s', label = the token at the cut) of F_cut(s) w B_behind(s'): every
// path crosses the boundary between two columns exactly once (a sum over states of one column would count a path once per
// state it visits there through silent moves).  One workgroup per sequence, exact fp64 log / exp.
__global__ __launch_bounds__(256) void k_onetape_join(DevMachine m, const PairDesc *__restrict__ pairs, int inputTape, const int *__restrict__ tape,
                                                      const double *__restrict__ fvec, const double *__restrict__ bvec, double *__restrict__ loglike) {
  __shared__ double red[256];
  const long long p = blockIdx.x;
  const PairDesc pd = pairs[p];
  const int S = m.S, mid = inputTape ? pd.inLen : pd.outLen;      // (pairs: the PREFIXES' descriptors -- the cut sits behind a prefix's last symbol)
  const int y = tape[(inputTape ? pd.inBase : pd.outBase) + mid];   // (the key of a one-tape label is the token itself)
  const double *F = fvec + p * (long long)S, *B = bvec + p * (long long)S;
  double mx = -INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) {
    const double f = F[s];
    const int row = s * m.K + y;
    for (int a = m.outOff[row]; a < m.outOff[row + 1]; ++a) mx = fmax(mx, f + (m.outW[a] + B[m.outDst[a]]));
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if ((int)threadIdx.x < h) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + h]); __syncthreads(); }
  mx = red[0];
  __syncthreads();
  double sum = 0.0;
  if (mx > -INFINITY)
    for (int s = threadIdx.x; s < S; s += 256) {
      const double f = F[s];
      const int row = s * m.K + y;
      for (int a = m.outOff[row]; a < m.outOff[row + 1]; ++a) sum += exp(f + (m.outW[a] + B[m.outDst[a]]) - mx);
    }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
  if (threadIdx.x == 0) loglike[p] = mx > -INFINITY ? mx + log(red[0]) : -INFINITY;
}

int wide_join(const mb_machine *m, const PairDesc *d_pairs, long long nPairs, const int *d_tape, const double *fvec, const double *bvec,
              double *loglike, hipStream_t st) {
  if (nPairs <= 0) return 0;
  hipLaunchKernelGGL(k_onetape_join, dim3((unsigned)nPairs), dim3(256), 0, st, m->dev, d_pairs, m->nIn != 0 ? 1 : 0, d_tape, fvec, bvec, loglike);
  return hip_ok(hipGetLastError(), "one-tape join launch") ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------
// posterior transition counts of one-tape machines (BackwardMatrix::getCounts, src/backward.cpp:58-87)
// ------------------------------------------------------------------------------------------------------------
// count[e] += sum over columns of exp(F(src) - LL + B(dst) + w) has no dependency between columns, so it is not a sweep: a
// LANE OWNS ONE TRANSITION and walks the columns of a sequence, summing in a register -- one atomic per (transition,
// sequence part) instead of one per candidate.  Transitions are sorted by label, so a wavefront's 64 transitions share it
// and a column whose token differs is skipped by a scalar branch.  The 2 x nStates doubles of a column are gathered by
// every edge block of the sequence; blockIdx -> (sequence part, edge block) keeps the edge blocks of one part on ONE XCD
// and next to each other in launch order, so they walk the columns together and the gathers hit that XCD's L2.
struct OtEdge { uint32_t src, pos; };
// one atomic per (transition, sequence part): fp64, or 64-bit fixed point at 2^-36 (deterministic mode, mb_internal.h)
__device__ __forceinline__ void count_add(double *counts, uint32_t e, double x, int det) {
  if (det) atomicAdd((unsigned long long *)counts + e, (unsigned long long)fmin(fmax(x * 68719476736.0 + 0.5, 0.0), 4611686018427387904.0));      // saturates (NaN -> 0): the host reads >= 2^62 as overflow
  else atomicAdd(&counts[e], x);
}          // pos: position in the outgoing view (weight, destination, edge id); ~0u = padding

__global__ __launch_bounds__(256) void k_onetape_counts(DevMachine m, const OtEdge *__restrict__ edges,
                                                        const unsigned short *__restrict__ waveLabel, int nWaves, int nEB,
                                                        const PairDesc *__restrict__ pairs, long long nUnits, int colSplit, int inputTape,
                                                        const int *__restrict__ tape, const double *__restrict__ fwd,
                                                        const double *__restrict__ bwd, double *__restrict__ counts, const float *__restrict__ norm, int det) {
  const long long bid = blockIdx.x, slot = bid >> 3;
  const long long unit = (slot / nEB) * 8 + (bid & 7);
  const int eb = (int)(slot % nEB);
  if (unit >= nUnits) return;
  const long long p = unit / colSplit;
  const int part = (int)(unit - p * colSplit);
  const PairDesc pd = pairs[p];
  const int L = inputTape ? pd.inLen : pd.outLen;
  const long long S = m.S;
  const double *F = fwd + pd.cellBase, *B = bwd + pd.cellBase;
  const double ll = B[0];                       // BackwardMatrix::logLike() = cell(0,0,start), src/backward.cpp:48-50,66
  if (!(ll > -INFINITY)) return;
  const int wave = __builtin_amdgcn_readfirstlane(eb * 4 + (int)(threadIdx.x >> 6));
  if (wave >= nWaves) return;
  const int y = waveLabel[wave];
  const OtEdge e = edges[(long long)wave * 64 + (threadIdx.x & 63)];
  const bool live = e.pos != 0xFFFFFFFFu;
  const uint32_t pos = live ? e.pos : 0u;
  const long long src = live ? e.src : 0, dst = m.outDst[pos];
  const double wl = live ? m.outW[pos] - ll : -INFINITY;
  const int per = (L + colSplit) / colSplit, cA = part * per, cB = min(L + 1, cA + per);   // source columns [cA, cB)
  const int *tk = tape + (inputTape ? pd.inBase : pd.outBase);
  auto term = [&](long long cs, long long cd) -> double {
    return (double)__builtin_amdgcn_exp2f((float)(F[cs * S + src] + (B[cd * S + dst] + wl)) * 1.44269504088896f);
  };
  // norm: one over what the emitting terms of each column sum to (k_onetape_colnorm; see k_onetape_counts_lds), by global column
  const float *nm = norm ? norm + pd.cellBase / S : nullptr;
  double acc = 0.0;
  if (y == 0) {
#pragma unroll 4
    for (int c = cA; c < cB; ++c) acc += nm ? term(c, c) * (double)nm[c] : term(c, c);
  } else {
    const int cE = min(cB, L);
    for (int c = cA; c < cE; ++c)
      if (tk[c] == y) acc += nm ? term(c, c + 1) * (double)nm[c] : term(c, c + 1);
  }
  if (live && acc != 0.0) count_add(counts, m.outEid[pos], acc, det);
}

// norm[global column] = 1 / (sum of the emitting terms that leave the column), the last column of a sequence: 1 / exp(F(L, end) -
// logLike) -- the normaliser of k_onetape_counts_lds for the kernel above, whose edge blocks are workgroups of their own.  One
// workgroup per sequence part, thread = source state (the outgoing view: row s * K + token).
__global__ __launch_bounds__(256) void k_onetape_colnorm(DevMachine m, const PairDesc *__restrict__ pairs, int colSplit, int inputTape,
                                                         const int *__restrict__ tape, const double *__restrict__ fwd,
                                                         const double *__restrict__ bwd, float *__restrict__ norm) {
  __shared__ float red[4];
  const long long unit = blockIdx.x, p = unit / colSplit;
  const int part = (int)(unit - p * colSplit), tid = threadIdx.x;
  const PairDesc pd = pairs[p];
  const int L = inputTape ? pd.inLen : pd.outLen;
  const long long S = m.S;
  const double *F = fwd + pd.cellBase, *B = bwd + pd.cellBase;
  float *nm = norm + pd.cellBase / S;
  const double ll = B[0];
  const int per = (L + colSplit) / colSplit, cA = part * per, cB = min(L + 1, cA + per);
  const int *tk = tape + (inputTape ? pd.inBase : pd.outBase);
  for (int c = cA; c < cB; ++c) {
    float z = 0.0f;
    if (!(ll > -INFINITY)) z = 1.0f;
    else if (c == L) z = __builtin_amdgcn_exp2f((float)(F[(long long)L * S + (S - 1)] - ll) * 1.44269504088896f);
    else {
      const int y = tk[c];
      const double *Fc = F + (long long)c * S, *Bn = B + (long long)(c + 1) * S;
      for (int s = tid; s < (int)S; s += 256) {
        const double f = Fc[s] - ll;
        const int row = s * m.K + y;
        for (int a = m.outOff[row]; a < m.outOff[row + 1]; ++a) z += __builtin_amdgcn_exp2f((float)(f + (m.outW[a] + Bn[m.outDst[a]])) * 1.44269504088896f);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off);
      __syncthreads();                                    // (the previous column's sums have been read)
      if ((tid & 63) == 0) red[tid >> 6] = z;
      __syncthreads();
      z = red[0] + red[1] + red[2] + red[3];
    }
    if (tid == 0) nm[c] = (z > 0.0f && z < 3.0e38f) ? 1.0f / z : 1.0f;
  }
}

// The same sum with the columns staged in LDS, for machines whose three state vectors F(c), B(c), B(c + 1) fit it (<= 6 800
// states: the 20-node config-5 machine): one workgroup per sequence part walks its columns from the back, loads F(c) and B(c)
// with coalesced reads (B(c + 1) is the previous iteration's B(c)), and every thread evaluates its EPT transitions from LDS.
// HBM traffic = both matrices exactly once, no L2 gathers (the kernel above: 12 ms on 64 x 2 kb, this one: the time of
// streaming 10.4 GB).
// NL: loads per thread and state vector of a column (ceil(S / 512)): the staging loop is unrolled so that a thread's 2 NL loads of a
// column are ALL in flight before the first one is stored to LDS.  As a run-time loop every iteration waited for its own two loads --
// ten dependent trips to HBM per column on config 5's machine, 11.4 us per column where the compute phase needs 2 (71 ms per 32 sequences
// x 50 kb = 1.8 TB/s; profiles/r06_onetape_estep_kernel_stats.csv).
template <int EPT, int NL>
__global__ __launch_bounds__(512) void k_onetape_counts_lds(DevMachine m, const OtEdge *__restrict__ edges, const unsigned short *__restrict__ waveLabel,
                                                            int nEdges, const PairDesc *__restrict__ pairs, int colSplit, int inputTape,
                                                            const int *__restrict__ tape, const double *__restrict__ fwd,
                                                            const double *__restrict__ bwd, double *__restrict__ counts, int normalise, int det) {
  extern __shared__ double cl[];
  const int S = m.S, tid = threadIdx.x;
  double *Fl = cl, *Ba = cl + S, *Bb = cl + 2 * (long long)S;
  float *zred = (float *)(cl + 3 * (long long)S);   // [16] per-wavefront sums of a column's emitting terms, two columns alternating
  const long long unit = blockIdx.x, p = unit / colSplit;
  const int part = (int)(unit - p * colSplit);
  const PairDesc pd = pairs[p];
  const int L = inputTape ? pd.inLen : pd.outLen;
  const double *F = fwd + pd.cellBase, *B = bwd + pd.cellBase;
  const double ll = B[0];                       // BackwardMatrix::logLike() = cell(0,0,start), src/backward.cpp:48-50,66
  if (!(ll > -INFINITY)) return;
  // source | destination << 16 (S < 65 536 here), label, weight x log2(e) -- fp32: the term's exponent (F - LL) + B is formed in fp64 and
  // rounded to fp32 anyway, the weight joins it in the conversion to base 2 --, usage.  Four registers a transition: with the weight
  // in fp64 the 32-transition form spilled (17 VGPRs, 72 bytes of scratch per lane)
  uint32_t esd[EPT]; int elab[EPT]; float ewl[EPT]; double acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + k * 512;
    const bool in = e < nEdges;
    const OtEdge ed = in ? edges[e] : OtEdge{0u, 0xFFFFFFFFu};
    const bool live = ed.pos != 0xFFFFFFFFu;
    esd[k] = live ? ((ed.src << 3) | (m.outDst[ed.pos] << 19)) : 0u;      // BYTE offsets into a staged vector (three vectors fit the LDS: S < 8 192)
    ewl[k] = live ? (float)(m.outW[ed.pos] * 1.4426950408889634) : -INFINITY;
    // (the edge list is padded to whole wavefronts per label: the label is uniform inside a wavefront -> scalar registers)
    elab[k] = __builtin_amdgcn_readfirstlane((tid & ~63) + k * 512 < nEdges ? (int)waveLabel[((tid & ~63) + k * 512) >> 6] : -1);
    acc[k] = 0.0;
  }
  const int per = (L + colSplit) / colSplit, cA = part * per, cB = min(L + 1, cA + per);   // source columns [cA, cB)
  const int *tk = tape + (inputTape ? pd.inBase : pd.outBase);
  double *Bcur = Ba, *Bnext = Bb;
  if (cB <= L) for (int j = tid; j < S; j += 512) Bcur[j] = B[(long long)cB * S + j];   // B(cB): "next" of the first column handled
  for (int c = cB - 1; c >= cA; --c) {
    { double *t = Bcur; Bcur = Bnext; Bnext = t; }                // Bnext = B(c + 1), Bcur is refilled with B(c)
    __syncthreads();                                              // the previous column's reads are done
    constexpr int NLB = (EPT >= 32 && NL > 10) ? (NL + 1) / 2 : NL;      // (32 transitions a thread leave registers for 20 loads in flight, not 28)
#pragma unroll
    for (int q0 = 0; q0 < NL; q0 += NLB) {
      double fr[NLB], br[NLB];
      const double *Fc = F + (long long)c * S, *Bc = B + (long long)c * S;
#pragma unroll
      for (int q = 0; q < NLB; ++q) { const int j = tid + (q0 + q) * 512; if (q0 + q < NL && j < S) { fr[q] = Fc[j]; br[q] = Bc[j]; } }
#pragma unroll
      for (int q = 0; q < NLB; ++q) { const int j = tid + (q0 + q) * 512; if (q0 + q < NL && j < S) { Fl[j] = fr[q] - ll; Bcur[j] = br[q]; } }      // (F - LL: every term's exponent needs it)
    }
    __syncthreads();
    const int y = c < L ? tk[c] : -2;
    // The terms of this column, and their NORMALISER.  In exact arithmetic the emitting transitions that leave column c sum to 1
    // (every path crosses the boundary c | c + 1 exactly once: k_onetape_join) -- in the sweeps' arithmetic F(c, .) carries the
    // rounding of c columns of log-sum-exp and B(c, .) of L - c, a random walk of ~1e-7 steps that reached 2e-4 per transition
    // on 50 000 columns (test_baseline_config5_one_sequence_at_50kb_against_the_oracle) while the RATIOS inside a column stay
    // good to 1e-7.  So every term of column c is divided by what its emitting terms sum to; the last column, which no
    // transition leaves, by exp(F(L, end) - logLike), the same drift read off the end cell.
    float t[EPT];
    float zpart = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      t[k] = 0.0f;
      // (the packed offsets stay packed: unpacked once outside the column loop they would cost two registers a transition instead of one)
      uint32_t sd = esd[k];
      asm volatile("" : "+v"(sd));
      const double fs = *(const double *)((const char *)Fl + (sd & 0xffffu));
      if (elab[k] == 0) t[k] = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(fs + *(const double *)((const char *)Bcur + (sd >> 16))), 1.44269504088896f, ewl[k]));
      else if (elab[k] == y) { t[k] = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(fs + *(const double *)((const char *)Bnext + (sd >> 16))), 1.44269504088896f, ewl[k])); zpart += t[k]; }
    }
    float inv;
    if (normalise) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) zpart += __shfl_xor(zpart, off);
      if ((tid & 63) == 0) zred[(c & 1) * 8 + (tid >> 6)] = zpart;
      __syncthreads();
      float z = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) z += zred[(c & 1) * 8 + w];
      if (c == L) z = __builtin_amdgcn_exp2f((float)Fl[S - 1] * 1.44269504088896f);
      inv = (z > 0.0f && z < 3.0e38f) ? 1.0f / z : 1.0f;
    } else inv = 1.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) acc[k] += (double)(t[k] * inv);
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + k * 512;
    if (e < nEdges && acc[k] != 0.0) { const uint32_t pos = edges[e].pos; if (pos != 0xFFFFFFFFu) count_add(counts, m.outEid[pos], acc[k], det); }
  }
}

template <int EPT, int NL>
static void launch_counts_lds(const mb_machine *m, const WideCountPlan &C, const PairDesc *d_desc, long long nUnits, int colSplit, bool inputTape,
                              const int *d_tape, const double *fwd, const double *bwd, double *d_counts, hipStream_t st) {
  const size_t lds = (size_t)3 * m->S * sizeof(double) + 16 * sizeof(float);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)&k_onetape_counts_lds<EPT, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((k_onetape_counts_lds<EPT, NL>), dim3((unsigned)nUnits), dim3(512), lds, st, m->dev, (const OtEdge *)C.d_edges,
                     (const unsigned short *)C.d_waveLabel, C.nWaves * 64, d_desc, colSplit, inputTape ? 1 : 0, d_tape, fwd, bwd, d_counts,
                     env_int_w("MB_ONETAPE_COUNT_NORMALISE", 1), g_deterministic ? 1 : 0);
}

bool wide_counts_build(const mb_machine *m, WideCountPlan &C) {
  if ((m->nIn != 0) == (m->nOut != 0)) return false;
  const bool inputTape = m->nIn != 0;
  std::vector<uint32_t> posOf(m->nTrans);
  for (long long a = 0; a < m->nTrans; ++a) posOf[m->outPerm[a]] = (uint32_t)a;
  auto label = [&](long long e) { return (int)(inputTape ? m->inTok[e] : m->outTok[e]); };
  std::vector<uint32_t> ids(m->nTrans);
  std::iota(ids.begin(), ids.end(), 0u);
  std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) {
    if (label(a) != label(b)) return label(a) < label(b);
    if (m->src[a] != m->src[b]) return m->src[a] < m->src[b];
    return m->dst[a] < m->dst[b];
  });
  std::vector<OtEdge> edges;
  std::vector<unsigned short> wl;
  for (size_t k = 0; k < ids.size();) {
    const int y = label(ids[k]);
    size_t k1 = k;
    while (k1 < ids.size() && label(ids[k1]) == y) ++k1;
    for (size_t j = k; j < k1; ++j) edges.push_back({m->src[ids[j]], posOf[ids[j]]});
    while (edges.size() % 64) edges.push_back({0u, 0xFFFFFFFFu});
    wl.resize(edges.size() / 64, (unsigned short)y);
    k = k1;
  }
  C.nWaves = (int)wl.size();
  if (C.d_edges) (void)hipFree(C.d_edges);
  if (C.d_waveLabel) (void)hipFree(C.d_waveLabel);
  C.d_edges = nullptr; C.d_waveLabel = nullptr;
  if (!hip_ok(hipMalloc(&C.d_edges, std::max<size_t>(edges.size(), 1) * sizeof(OtEdge)), "hipMalloc(count edges)") ||
      !hip_ok(hipMalloc(&C.d_waveLabel, std::max<size_t>(wl.size(), 1) * sizeof(unsigned short)), "hipMalloc(count labels)")) return false;
  if (!edges.empty() && (!hip_ok(hipMemcpy(C.d_edges, edges.data(), edges.size() * sizeof(OtEdge), hipMemcpyHostToDevice), "H2D(count edges)") ||
                         !hip_ok(hipMemcpy(C.d_waveLabel, wl.data(), wl.size() * sizeof(unsigned short), hipMemcpyHostToDevice), "H2D(count labels)"))) return false;
  C.ok = true;
  return true;
}

void wide_counts_free(WideCountPlan &C) {
  if (C.d_edges) (void)hipFree(C.d_edges);
  if (C.d_waveLabel) (void)hipFree(C.d_waveLabel);
  C = WideCountPlan();
}

int wide_counts(const mb_machine *m, const WideCountPlan &C, const PairDesc *d_desc, const std::vector<PairDesc> &hp, const int *d_tape,
                const double *fwd, const double *bwd, double *d_counts, hipStream_t st) {
  if (!C.ok) { set_error("one-tape count plan not built"); return 1; }
  if (hp.empty() || C.nWaves == 0) return 0;
  const bool inputTape = m->nIn != 0;
  // few sequences: cut them into column parts so that every XCD has work (a part is never shorter than 64 columns)
  int maxLen = 0;
  for (const PairDesc &pd : hp) maxLen = std::max(maxLen, inputTape ? pd.inLen : pd.outLen);
  int colSplit = 1;
  const int want = env_int_w("MB_ONETAPE_COUNT_UNITS", 32);
  while ((long long)hp.size() * colSplit < want && (maxLen + 1) / (colSplit * 2) >= 64) colSplit *= 2;
  // columns staged in LDS when three state vectors fit and a thread's share of the transitions fits its registers
  const long long nE = (long long)C.nWaves * 64;
  if ((size_t)3 * m->S * sizeof(double) <= 158 * 1024 && nE <= 32 * 512 && env_int_w("MB_ONETAPE_COUNTS_LDS", 1)) {
    // one workgroup per CU (three state vectors in LDS): the sequences are cut into column parts (>= 64 columns) until the launch fills
    // the chip AND its last round of workgroups is nearly full -- 33 sequences x 8 parts = 264 workgroups on 256 CUs ran two rounds for
    // the work of one (141 ms where 31 x 16 parts took 71; profiles/r06_onetape_estep_trace.txt)
    int cs = 1;
    const int wantUnits = env_int_w("MB_ONETAPE_COUNT_LDS_UNITS", 256);
    int cus = 0;
    { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; (void)hipGetLastError(); }
    auto tailOk = [&](long long units) { if (cus <= 0) return true; const long long rounds = (units + cus - 1) / cus; return (double)units >= 0.9 * (double)(rounds * cus); };
    while (((long long)hp.size() * cs < wantUnits || !tailOk((long long)hp.size() * cs)) && (maxLen + 1) / (cs * 2) >= 64 && cs < 1024) cs *= 2;
    const long long units = (long long)hp.size() * cs;
    const int nl = (int)((m->S + 511) / 512);      // <= 14: three state vectors fit the LDS
#define OT_COUNTS_GO(E) (nl <= 4 ? launch_counts_lds<E, 4>(m, C, d_desc, units, cs, inputTape, d_tape, fwd, bwd, d_counts, st) \
                         : nl <= 8 ? launch_counts_lds<E, 8>(m, C, d_desc, units, cs, inputTape, d_tape, fwd, bwd, d_counts, st) \
                         : nl <= 10 ? launch_counts_lds<E, 10>(m, C, d_desc, units, cs, inputTape, d_tape, fwd, bwd, d_counts, st) \
                         : launch_counts_lds<E, 14>(m, C, d_desc, units, cs, inputTape, d_tape, fwd, bwd, d_counts, st))
    if (nE <= 8 * 512) OT_COUNTS_GO(8);
    else if (nE <= 16 * 512) OT_COUNTS_GO(16);
    else OT_COUNTS_GO(32);
#undef OT_COUNTS_GO
    return hip_ok(hipGetLastError(), "one-tape counts launch") ? 0 : 1;
  }
  const long long nUnits = (long long)hp.size() * colSplit;
  const int nEB = (C.nWaves + 3) / 4;
  const long long grid = ((nUnits + 7) / 8) * 8 * nEB;
  if (grid > 0x7fffffffll) { set_error("one-tape counts: launch too large"); return 1; }
  float *d_norm = nullptr;
  if (env_int_w("MB_ONETAPE_COUNT_NORMALISE", 1)) {
    long long cols = 0;
    for (const PairDesc &pd : hp) cols = std::max(cols, pd.cellBase / std::max(m->S, 1) + (inputTape ? pd.inLen : pd.outLen) + 1);
    d_norm = (float *)ws_get(13, (size_t)std::max<long long>(cols, 1) * sizeof(float));
    if (!d_norm) return 1;
    int cs = 1;
    while ((long long)hp.size() * cs < 1024 && (maxLen + 1) / (cs * 2) >= 16) cs *= 2;
    hipLaunchKernelGGL(k_onetape_colnorm, dim3((unsigned)(hp.size() * cs)), dim3(256), 0, st, m->dev, d_desc, cs, inputTape ? 1 : 0, d_tape, fwd, bwd, d_norm);
  }
  hipLaunchKernelGGL(k_onetape_counts, dim3((unsigned)grid), dim3(256), 0, st, m->dev, (const OtEdge *)C.d_edges,
                     (const unsigned short *)C.d_waveLabel, C.nWaves, nEB, d_desc, nUnits, colSplit, inputTape ? 1 : 0, d_tape, fwd, bwd, d_counts, (const float *)d_norm, g_deterministic ? 1 : 0);
  return hip_ok(hipGetLastError(), "one-tape counts launch") ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------
// Viterbi traceback of a one-tape machine
// ------------------------------------------------------------------------------------------------------------
// The generic walker (mb_generic.hip k_traceback) pays four to five dependent trips to memory per path step (CSR row offsets ->
// edges -> cells -> edge id -> labels): 1.6 us a step, 6.4 ms for the 4 000 steps of a 2 kb sequence on the 20-node machine --
// 60 % on top of the retimed fill.  Here one workgroup owns a sequence: the column of the position and the one before it --
// every cell a step can read -- sit in LDS together with one offset per state, fifteen wavefronts fetch the columns ahead
// while the first one walks, and a state's incoming edges (contiguous in the `incoming` view; id, source and label packed
// in 8 bytes, weight beside them) come in ONE coalesced trip that hits the L2; the maximum is a butterfly over the values
// alone and the winner the first lane that holds it.  Candidate order and tie-break are the reference's
// (src/dpmatrix.defs.h:82-110): the emitting group before the silent one, position inside the group, first maximum wins.
// The walk itself is ~0.3 us a step now; what is left is the re-read of the whole matrix, column by column (8 B per cell at
// what 64 CUs keep in flight: 1.4 TB/s): 3.7 ms instead of 6.4 -- fill + paths 17.3 -> 14.5 ms at 64 x 2 kb.  One traceback
// code per cell written by the fill would remove that read.
struct OtTbEdge { uint32_t eid; uint16_t src; uint16_t key; };      // key: the token the transition emits (0: silent)

__global__ __launch_bounds__(1024) void k_onetape_traceback(DevMachine m, const OtTbEdge *__restrict__ edges, const int *__restrict__ begin,
                                                           const PairDesc *__restrict__ pairs, const int *__restrict__ tape,
                                                           const double *__restrict__ pool, const long long *__restrict__ slotOff,
                                                           uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  extern __shared__ double tlds[];
  const int S = m.S, tid = threadIdx.x, lane = tid & 63;
  double *col = tlds;                                    // [3][S]: column c in buffer c % 3
  int *sBeg = (int *)(tlds + 3 * (size_t)S);             // [S + 1]
  int &done = sBeg[S + 1];                               // 1: the walk reached the start, 2: it failed (pathLen says how)
  const long long p = blockIdx.x;
  const PairDesc pd = pairs[p];
  const bool inputTape = m.nOut == 0;
  const int L = inputTape ? pd.inLen : pd.outLen;
  const int *tok = tape + (inputTape ? pd.inBase : pd.outBase);
  const double *cells = pool + pd.cellBase;              // column c at cells + c * S (the other tape is empty)
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  for (int k = tid; k <= S; k += 1024) sBeg[k] = begin[k];
  for (int k = tid; k < S; k += 1024) {
    col[(L % 3) * S + k] = cells[(long long)L * S + k];
    if (L >= 1) col[((L - 1) % 3) * S + k] = cells[(long long)(L - 1) * S + k];
  }
  if (tid == 0) done = 0;
  __syncthreads();
  if (tid == 0 && !(col[(L % 3) * S + S - 1] > -INFINITY)) { pathLen[p] = -1; done = 2; }
  __syncthreads();
  int s = S - 1;
  long long n = 0;
  uint32_t held = 0;
  typedef volatile __attribute__((address_space(3))) int lds_flag;      // (a read through a generic pointer would be a FLAT load)
  lds_flag *doneFlag = (lds_flag *)(uintptr_t)(unsigned)(uintptr_t)&done;
  // A column fresh from the fill comes from HBM, and 40 KB per column and workgroup is bound by what a CU keeps in flight
  // (one batch of loads per round trip: 1.4 us, twice the walker's two steps): the loaders therefore run TWO columns ahead -- the
  // loads of column o - 3 are issued before the barrier of iteration o and stay in flight across it (the barrier below waits
  // for LDS traffic only), their values go to LDS one iteration later.
  typedef double pair_t __attribute__((ext_vector_type(2)));
  typedef pair_t pair_u __attribute__((aligned(8)));                     // 16-byte loads at 8-byte alignment (S may be odd)
  struct Pair { double a, b; };
  Pair r[4];
  auto fetch = [&](int c) {
    const double *src = cells + (long long)c * S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 2 * (tid - 64) + j * 1920;
      if (c >= 0 && k + 1 < S) { const pair_t v = __builtin_nontemporal_load((const pair_u *)(src + k)); r[j] = Pair{v.x, v.y}; }
      else r[j] = Pair{(c >= 0 && k < S) ? __builtin_nontemporal_load(src + k) : 0.0, 0.0};
    }
  };
  if (tid >= 64) fetch(L - 2);
  for (int o = L; !*doneFlag; --o) {
    if (tid >= 64) {
      // the column two back, for the steps after the next emission
      if (o >= 2) {
        double *dst = col + ((o - 2) % 3) * S;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = 2 * (tid - 64) + j * 1920; if (k < S) dst[k] = r[j].a; if (k + 1 < S) dst[k + 1] = r[j].b; }
      }
      fetch(o - 3);
    } else {
      const double *cur = col + (o % 3) * S, *prev = col + ((o + 2) % 3) * S;      // (o - 1) % 3
      const int ot = o ? tok[o - 1] : 0;
      for (;;) {
        if (o == 0 && s == 0) { if (lane == 0) done = 1; break; }
        int bestA = -1;
        uint32_t bestEid = 0, bestSk = 0;                  // the winning edge travels with the maximum: no second trip for it
        const int b0 = sBeg[s], b1 = sBeg[s + 1];
        if (b1 - b0 <= 64) {
          // the usual case, one candidate per lane: maximum by a butterfly over the values alone, then the first lane that
          // holds it -- emitting group before silent; inside a group the edge list, hence the lanes, are in the reference's order
          const int a = b0 + lane;
          const bool valid = a < b1;
          OtTbEdge e{0u, 0, 0};
          double w = 0.0;
          if (valid) { e = edges[a]; w = m.inW[a]; }
          const int key = (int)e.key;
          const bool emits = valid && o && key != 0 && key == ot, has = emits || (valid && key == 0);
          // (a silent self-loop on state 0 is a genuine candidate of the reference's traceback: kept, as in k_traceback)
          const double v = has ? (emits ? prev : cur)[e.src] + w : -INFINITY;
          double mx = v;
          wide_max_all<MB_VITERBI, 1>(mx); wide_max_all<MB_VITERBI, 2>(mx); wide_max_all<MB_VITERBI, 4>(mx);
          wide_max_all<MB_VITERBI, 8>(mx); wide_max_all<MB_VITERBI, 16>(mx); wide_max_all<MB_VITERBI, 32>(mx);
          const unsigned long long eqE = __ballot(emits && v == mx), eqS = __ballot(has && !emits && v == mx);
          if (eqE | eqS) {
            const int win = eqE ? __ffsll((long long)eqE) - 1 : __ffsll((long long)eqS) - 1;
            bestA = b0 + win;
            bestEid = (uint32_t)__builtin_amdgcn_readlane((int)e.eid, win);
            bestSk = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)e.src | ((uint32_t)e.key << 16)), win);
          }
        } else {
          double best = -INFINITY; int bestIdx = 0x7fffffff;
          for (int a = b0 + lane; a < b1; a += 64) {
            const OtTbEdge e = edges[a];
            const int key = (int)e.key;
            int grp; const double *sc;
            if (key == 0) { grp = 1; sc = cur; }
            else if (o && key == ot) { grp = 0; sc = prev; }
            else continue;
            const double v = sc[e.src] + m.inW[a];
            const int idx = (grp << 24) | (a - b0);
            if (bestA < 0 || v > best || (v == best && idx < bestIdx)) { best = v; bestIdx = idx; bestA = a; bestEid = e.eid; bestSk = (uint32_t)e.src | ((uint32_t)e.key << 16); }
          }
          for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bestIdx, off), oa = __shfl_xor(bestA, off);
            const uint32_t oe = (uint32_t)__shfl_xor((int)bestEid, off), ok = (uint32_t)__shfl_xor((int)bestSk, off);
            const bool take = oa >= 0 && (bestA < 0 || ov > best || (ov == best && oi < bestIdx));
            if (take) { best = ov; bestIdx = oi; bestA = oa; bestEid = oe; bestSk = ok; }
          }
        }
        if (bestA < 0) { if (lane == 0) { pathLen[p] = -3; done = 2; } break; }
        if (n >= cap) { if (lane == 0) { pathLen[p] = -2; done = 2; } break; }
        const OtTbEdge be{bestEid, (uint16_t)(bestSk & 0xffffu), (uint16_t)(bestSk >> 16)};
        if ((int)(n & 63) == lane) held = be.eid;      // one store per 64 steps (mb_generic.hip k_traceback)
        ++n;
        if ((n & 63) == 0) pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = held;
        s = (int)be.src;
        if (be.key) break;                              // an emission: the position moves one column back
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS traffic only: the loaders' next column stays in flight
  }
  if (tid < 64 && done == 1) {
    if ((n & ~63ll) + lane < n) pathBuf[slot0 + cap - 1 - ((n & ~63ll) + lane)] = held;
    if (lane == 0) pathLen[p] = n;
  }
}

bool wide_traceback_build(const mb_machine *m, WideTbPlan &T) {
  T.tried = true; T.ok = false;
  if ((m->nIn != 0) == (m->nOut != 0) || !env_int_w("MB_ONETAPE_TRACEBACK", 1)) return false;
  const int S = m->S, K = (m->nIn + 1) * (m->nOut + 1);
  T.ldsBytes = 3 * (size_t)S * sizeof(double) + (size_t)(S + 2) * sizeof(int);
  if (T.ldsBytes > WIDE_LDS_MAX - 64 || S > 65535 || K > 65535) return false;
  const bool inputTape = m->nOut == 0;
  std::vector<OtTbEdge> edges((size_t)m->nTrans);
  for (long long a = 0; a < m->nTrans; ++a) {
    const uint32_t e = m->inPerm[a];
    edges[a] = OtTbEdge{e, (uint16_t)m->src[e], (uint16_t)(inputTape ? m->inTok[e] : m->outTok[e])};
  }
  std::vector<int> begin((size_t)S + 1);
  for (int st = 0; st <= S; ++st) begin[st] = m->inOff[(size_t)st * K];
  if (!hip_ok(hipMalloc(&T.d_edges, std::max<size_t>(edges.size(), 1) * sizeof(OtTbEdge)), "hipMalloc(traceback edges)") ||
      !hip_ok(hipMalloc((void **)&T.d_begin, begin.size() * sizeof(int)), "hipMalloc(traceback offsets)")) return false;
  if ((!edges.empty() && !hip_ok(hipMemcpy(T.d_edges, edges.data(), edges.size() * sizeof(OtTbEdge), hipMemcpyHostToDevice), "H2D(traceback edges)")) ||
      !hip_ok(hipMemcpy(T.d_begin, begin.data(), begin.size() * sizeof(int), hipMemcpyHostToDevice), "H2D(traceback offsets)")) return false;
  T.ok = true;
  return true;
}

void wide_traceback_free(WideTbPlan &T) {
  if (T.d_edges) (void)hipFree(T.d_edges);
  if (T.d_begin) (void)hipFree(T.d_begin);
  T = WideTbPlan();
}

int wide_traceback(const mb_machine *m, const WideTbPlan &T, const PairDesc *d_pairs, long long nPairs, const int *d_tape, const double *d_pool,
                   const long long *d_slotOff, uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st) {
  if (nPairs <= 0) return 0;
  static bool attr = false;
  if (!attr) { MB_HIP(hipFuncSetAttribute((const void *)k_onetape_traceback, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX)); attr = true; }
  hipLaunchKernelGGL(k_onetape_traceback, dim3((unsigned)nPairs), dim3(1024), T.ldsBytes, st, m->dev, (const OtTbEdge *)T.d_edges, T.d_begin, d_pairs, d_tape,
                     d_pool, d_slotOff, d_pathBuf, d_pathLen);
  MB_HIP(hipGetLastError());
  return 0;
}

}  // namespace mb

// mb_medium.hip -- "lanes = states" tiled kernel family for medium-sized machines (17 <= nStates <= ~2000).
//
// Mapping (MI355X-first, not a translation of anything in the reference):
//   * a workgroup owns a STRIP of C consecutive input positions of one sequence pair and sweeps the output axis;
//     column c of the strip is one step behind column c-1, so the C supercells processed in one step lie on an
//     anti-diagonal of the lattice and are mutually independent;
//   * a wavefront processes G columns at once: its 64 lanes are split into G groups of LPG = 64/G lanes, and the
//     lanes of a group are the STATES of that column's supercell.  The states are scheduled in "rounds": every
//     round holds up to LPG states of one silent-transition level, so the silent chain inside a supercell
//     (src/forward.defs.h:32,43 relies on state order for it) costs one wave-local LDS round trip per level and
//     no workgroup barrier;
//   * the previous one/two anti-diagonals of the strip live in an LDS ring [slot][column][state] (fp64), so the
//     three neighbour supercells a cell reads -- (i-1,o-1), (i-1,o), (i,o-1) -- never touch HBM; the only HBM
//     traffic is the coalesced store of each finished supercell (materialised mode) and one halo supercell per
//     step from the strip to the left;
//   * the transition table is compiled on the host into per-round, per-token "slot" arrays laid out
//     [token][slot][lane] so that every lane's candidate (source state, log-weight) is a coalesced load.
//   * big lattices are cut into parallelogram tiles (TS steps of one strip); tile (strip a, block b) depends on
//     (a, b-1) and (a-1, b+1), so launch number 2a+b is a valid wavefront order and plain kernel boundaries are
//     the only inter-workgroup synchronisation (no spin-waits).
//   * Backward is the same kernel run on the reversed machine with reversed coordinates.
//
// Arithmetic: candidates cell+logW, the running maximum and all stored cells are fp64.  Viterbi uses max only and
// is bit-identical to the reference.  Forward keeps (max, sum of exp(cand-max)) with the sum in fp32 using
// v_exp_f32 / v_log_f32 (abs. error ~1e-7 per cell, far inside the 1e-4 relative tolerance); a state with a
// single candidate is exact.
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>

#include "mb_internal.h"
#include "mb_device_math.h"
#include "mb_medium.h"

namespace mb {

// ------------------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------------------
struct MedTileArgs {
  const PairDesc *pairs;
  const int *inTok, *outTok;
  double *pool;               // materialised matrices, reference layout; nullptr in rolling mode
  double *colHalo;            // rolling mode: per pair two [outLen+1][S] column buffers (ping-pong by strip parity)
  const long long *haloBase;  // rolling mode: per pair offset (in doubles) of its two buffers
  double *loglike;            // loglike[pairBase + blockIdx.y], written when the end cell is finalised (may be null)
  const int2 *tiles;          // materialised mode: (pair, strip) of workgroup tileBase + blockIdx.x
  int C, TS, launch, rev, materialise, tileBase, det;   // det: specialised kernel, count mode: accumulators in 64-bit fixed point (g_deterministic)
  const double *poolB;        // count mode (specialised kernel only): Backward matrices, same layout and cellBase as pool
  double *counts;             // count mode: [nTrans] posterior transition counts, accumulated with fp64 atomics
  const int *envStart, *envEnd;   // restricted envelopes (specialised kernel, JENV variant): rows at PairDesc::envBase
  // tiles without a matrix (specialised kernel, JMAT == 2): colHalo / haloBase then hold one halo column per strip
  double *bound;                  // tile-boundary records: the ring state a block hands to the next block of its strip
  const long long *boundBase;     // per pair offset (doubles)
  unsigned char *tb;              // Viterbi traceback bytes (MED_MODE_TB), medium_tb_stride(S) per supercell, PairDesc::cellBase = byte offset
};

#define MED_L2E 1.44269504088896f
#define MED_LN2 0.693147180559945f
constexpr int MS = MED_MAXSLOT;
constexpr int DW = MED_DESC_WORDS;

typedef const __attribute__((address_space(4))) int *cdesc_t;          // descriptors: constant address space -> s_load
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef const __attribute__((address_space(1))) u32x4 *grec_t;         // candidate records: global_load_dwordx4

struct Rec { double w; unsigned srcOff, dstOff; };

__device__ __forceinline__ Rec med_load(grec_t g, int idx) {
  const u32x4 r = g[idx];
  Rec o;
  o.w = __hiloint2double((int)r.y, (int)r.x);
  o.srcOff = r.z; o.dstOff = r.w;
  return o;
}

// first record of a chunk for this lane: offset + inTok*mulI + outTok*mulO + laneInGroup (24-bit multiplies: full
// rate); slot k of the chunk follows at k*stride.  All slots of a chunk belong to one table, hence read one vector.
__device__ __forceinline__ int med_idx0(cdesc_t dp, int it, int ot, int q) {
  return (int)__umul24(it, dp[2]) + (int)__umul24(ot, dp[3] & 0xFFFFFF) + dp[1] + q;
}

template <int N>
__device__ __forceinline__ void med_fetch(cdesc_t dp, grec_t g, int it, int ot, int q, Rec (&R)[MS]) {
  const int i0 = med_idx0(dp, it, ot, q), stride = dp[4];
#pragma unroll
  for (int k = 0; k < N; ++k) R[k] = med_load(g, i0 + k * stride);
}

// LDS byte offset of the vector a chunk reads, relative to the lane's own column: 0 diag, 1 left, 2 down, 3 cur.
__device__ __forceinline__ int med_vofs(int vsel, int sCur, int sPrev, int sPrev2, int colStride) {
  const int slotOff = vsel == 3 ? sCur : (vsel == 0 ? sPrev2 : sPrev);
  return slotOff - (vsel < 2 ? colStride : 0);
}

__device__ __forceinline__ double med_lds(const char *ldsb, int off) { return *(const double *)(ldsb + off); }

// pass 1 of a chunk: N candidates cell+logW (fp64) and their maximum
template <int N>
__device__ __forceinline__ void med_cands(const char *ldsb, int vecBase, const Rec (&R)[MS], double (&v)[MS], double &mx) {
  double x[N > 0 ? N : 1];
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = med_lds(ldsb, vecBase + (int)R[k].srcOff);
#pragma unroll
  for (int k = 0; k < N; ++k) { v[k] = x[k] + R[k].w; mx = dmax(mx, v[k]); }
}

template <int N>
__device__ __forceinline__ float med_sumexp(const double (&v)[MS], double gM) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) s += __builtin_amdgcn_exp2f((float)(v[k] - gM) * MED_L2E);
  return s;
}

static_assert(MED_MAXSLOT == 4, "MED_SWITCH enumerates the slot counts 1..MED_MAXSLOT");
#define MED_SWITCH(n, CALL)                                                                     \
  switch (n) {                                                                                  \
    case 1: { CALL(1); } break; case 2: { CALL(2); } break; case 3: { CALL(3); } break;         \
    default: { CALL(4); } break;                                                                \
  }

// workgroup barrier that orders LDS traffic only: global stores of finished supercells stay in flight across it
// (nothing in this launch reads them back), unlike __syncthreads() which drains vmcnt to zero every step.
__device__ __forceinline__ void med_block_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ void med_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- slow path: one supercell evaluated with generic loops (only the origin supercell of a pair uses it) ----------
template <int MODE>
__device__ __noinline__ void med_slow_supercell(cdesc_t desc, grec_t grec, int nChunks, const char *ldsb, int myColBase,
                                                int sCur, int sPrev, int sPrev2, int colStride, int it, int ot, int q,
                                                unsigned seedOff, bool origin, bool lanesOn) {
  double accM = -INFINITY; float accS = 0.0f;
  for (int ch = 0; ch < nChunks; ++ch) {
    cdesc_t dp = desc + ch * DW;
    const int hdr = dp[0];
    const int ns = hdr & 15;
    const bool first = (hdr >> 4) & 1, last = (hdr >> 5) & 1, sync = (hdr >> 6) & 1;
    unsigned dstOff = 0xFFFFFFFFu;
    const int vecBase = myColBase + med_vofs((unsigned)dp[3] >> 24, sCur, sPrev, sPrev2, colStride);
    for (int k = 0; k < ns; ++k) {
      const Rec r = med_load(grec, med_idx0(dp, it, ot, q) + k * dp[4]);
      if (k == 0) {
        dstOff = r.dstOff;
        if (first) {
          const bool seed = origin && dstOff == seedOff;   // cell(0,0,start) = 0 (src/forward.defs.h:36, viterbi.cpp:30)
          accM = seed ? 0.0 : -INFINITY; accS = seed ? 1.0f : 0.0f;
        }
      }
      const double v = med_lds(ldsb, vecBase + (int)r.srcOff) + r.w;
      if (MODE == MB_VITERBI) accM = dmax(accM, v);
      else {
        const double nm = dmax(accM, v), gM = (nm == -INFINITY) ? 0.0 : nm;
        accS = accS * __builtin_amdgcn_exp2f((float)(accM - gM) * MED_L2E) + __builtin_amdgcn_exp2f((float)(v - gM) * MED_L2E);
        accM = nm;
      }
    }
    if (last) {
      const double res = (MODE == MB_VITERBI) ? accM
                                               : ((accM == -INFINITY) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);
      if (lanesOn && (int)dstOff >= 0) *(double *)(ldsb + (myColBase + sCur + (int)dstOff)) = res;
    }
    if (sync) med_wave_sync();
  }
}

template <int MODE, int G>
__global__ __launch_bounds__(1024) void k_medium_tile(MedProgDev P, MedTileArgs A) {
  extern __shared__ double lds[];
  constexpr int LPG = 64 / G;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane / LPG, q = lane - g * LPG;
  const int S = P.S, Spad = P.Spad, NS = P.NS, C = A.C;
  // which tile.  Materialised launches enumerate exactly the live tiles in a 1-D grid: consecutive workgroup ids are
  // dealt round-robin to the 8 XCDs, so a dense list keeps every XCD at <= 32 resident workgroups (a 2-D grid with
  // early-exit holes put 33+ live tiles on some XCD and doubled the launch time).
  int pairIdx, a;
  if (A.materialise) { const int2 tl = A.tiles[A.tileBase + blockIdx.x]; pairIdx = tl.x; a = tl.y; }
  else { pairIdx = blockIdx.x; a = A.launch; }
  const PairDesc pd = A.pairs[pairIdx];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const long long I = inLen + 1;
  const int b = A.materialise ? A.launch - pd.launch0 - 2 * a : 0;
  const int NA = (inLen + C) / C;               // ceil((inLen+1)/C)
  const int T = outLen + C;                     // steps of one strip sweep: (outLen+1) + (C-1)
  if (a >= NA || b < 0 || (long long)b * A.TS >= T) return;
  const int t0 = b * A.TS, t1 = min(t0 + A.TS, T);
  const int i0 = a * C;
  const int c = wv * G + g;                      // my column
  const int i = i0 + c;
  const bool colValid = (c < C) && (i <= inLen);
  const int *in = A.inTok + pd.inBase, *out = A.outTok + pd.outBase;
  const int rev = A.rev;
  const int it = (colValid && i > 0) ? (rev ? in[inLen - i] : in[i - 1]) : 0;
  double *cells = A.materialise ? A.pool + pd.cellBase : nullptr;
  double *haloIn = nullptr, *haloOut = nullptr;
  if (!A.materialise) {
    double *hb = A.colHalo + A.haloBase[pairIdx];
    const long long hsz = (long long)(outLen + 1) * S;
    haloIn = hb + ((a + 1) & 1) * hsz;   // written by strip a-1
    haloOut = hb + (a & 1) * hsz;
  }
  // address of an already computed supercell (strip coordinates), S contiguous doubles
  auto cellPtr = [&](int ci, int co) -> double * {
    const long long ri = rev ? inLen - ci : ci, ro = rev ? outLen - co : co;
    return cells + (ro * I + ri) * S;
  };
  auto ring = [&](int slot, int col) -> double * { return lds + ((long long)slot * (C + 1) + col) * Spad; };

  // ---- LDS init: every vector starts as -inf, so that neighbours that do not exist (i = 0, o = 0), the sentinel
  //      entry [S] read by padding candidates, and not-yet-active columns all read -inf without any predicate ----
  for (int j = tid; j < NS * (C + 1) * Spad; j += blockDim.x) lds[j] = -INFINITY;
  __syncthreads();
  // ---- preload the ring state of steps t0-1 (and t0-2 when match edges exist) --------------------------------
  for (int dt = 1; dt < NS; ++dt) {
    const int tp = t0 - dt;
    const int slot = ((tp % NS) + NS) % NS;
    for (int idx = tid; idx < (C + 1) * S; idx += blockDim.x) {   // flat over (column, state)
      const int col = idx / S, j = idx - col * S;
      const int cc = col - 1, ci = i0 + cc, co = tp - cc;
      if (ci < 0 || ci > inLen || co < 0 || co > outLen) continue;
      const double *src = nullptr;
      if (A.materialise) src = cellPtr(ci, co);
      else if (cc == -1) src = haloIn + (long long)co * S;
      if (src) ring(slot, col)[j] = src[j];
    }
  }
  __syncthreads();

  cdesc_t desc = (cdesc_t)P.desc;
  grec_t grec = (grec_t)P.rec;
  const int nChunks = P.nChunks;
  const char *ldsb = (const char *)lds;
  const int colStride = Spad * 8, slotStride = (C + 1) * colStride;
  const int myColBase = (c + 1) * colStride;
  int slotCur = t0 % NS;
  int otNext = 0;
  {
    const int o = t0 - c;
    if (colValid && o > 0 && o <= outLen) otNext = rev ? out[outLen - o] : out[o - 1];
  }
  for (int t = t0; t < t1; ++t) {
    const int o = t - c;
    const bool active = colValid && o >= 0 && o <= outLen;
    const int ot = otNext;
    {  // prefetch the next step's output token
      const int on = o + 1;
      otNext = (colValid && on > 0 && on <= outLen) ? (rev ? out[outLen - on] : out[on - 1]) : 0;
    }
    const int slotPrev = (slotCur + NS - 1) % NS, slotPrev2 = (slotCur + NS - 2) % NS;
    const int sCur = slotCur * slotStride, sPrev = slotPrev * slotStride, sPrev2 = slotPrev2 * slotStride;
    // halo supercell (i0-1, t+1) for the next step, fetched cooperatively by the whole workgroup
    double hv[4];
    const bool wantHalo = (i0 > 0) && (t + 1 <= outLen);
    if (wantHalo) {
      const double *hs = A.materialise ? cellPtr(i0 - 1, t + 1) : haloIn + (long long)(t + 1) * S;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = tid + k * blockDim.x;
        hv[k] = (j < S) ? hs[j] : 0.0;
      }
    }
    if (t == 0 && a == 0) {
      // the origin supercell (0,0) is the only active one at step 0 of strip 0: generic slow path with the seed
      if (wv == 0) med_slow_supercell<MODE>(desc, grec, nChunks, ldsb, myColBase, sCur, sPrev, sPrev2, colStride, it, ot, q,
                                            P.seedOff, active && i == 0 && o == 0, active);
    } else {
      // ---- fast path: software pipeline over chunks; records of chunk ch+1 are fetched while ch is evaluated -----
      Rec RA[MS], RB[MS];
      double accM = -INFINITY; float accS = 0.0f;
      auto body = [&](int ch, Rec (&CUR)[MS], Rec (&NXT)[MS]) __attribute__((always_inline)) {
        cdesc_t dp = desc + ch * DW;
        const int hdr = dp[0];
        const int ns = hdr & 15, nsNext = (hdr >> 8) & 15;
        if (nsNext != 15) {   // 15 marks the last chunk
          cdesc_t dn = dp + DW;
#define CALL(N) med_fetch<N>(dn, grec, it, ot, q, NXT)
          MED_SWITCH(nsNext, CALL)
#undef CALL
        }
        const int vecBase = myColBase + med_vofs((unsigned)dp[3] >> 24, sCur, sPrev, sPrev2, colStride);
        if (hdr & (1 << 12)) {
          // every lane of this round has at most one candidate: the cell is exactly cand = source + logW
          const double res = med_lds(ldsb, vecBase + (int)CUR[0].srcOff) + CUR[0].w;
          if (active && (int)CUR[0].dstOff >= 0) *(double *)(ldsb + (myColBase + sCur + (int)CUR[0].dstOff)) = res;
        } else {
          const bool first = (hdr >> 4) & 1, last = (hdr >> 5) & 1;
          double v[MS];
          double mx = -INFINITY;
#define CALL(N) med_cands<N>(ldsb, vecBase, CUR, v, mx)
          MED_SWITCH(ns, CALL)
#undef CALL
          if (first) { accM = -INFINITY; accS = 0.0f; }
          if (MODE == MB_VITERBI) {
            accM = dmax(accM, mx);
          } else {
            // pass 2: sum of exp(candidate - max) in fp32
            const double newM = dmax(accM, mx);
            const double gM = (newM == -INFINITY) ? 0.0 : newM;
            float s = 0.0f;
#define CALL(N) s = med_sumexp<N>(v, gM)
            MED_SWITCH(ns, CALL)
#undef CALL
            if (!first) s += accS * __builtin_amdgcn_exp2f((float)(accM - gM) * MED_L2E);
            accS = s; accM = newM;
          }
          if (last) {
            double res;
            if (MODE == MB_VITERBI) res = accM;
            else res = ((accM == -INFINITY) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);
            if (active && (int)CUR[0].dstOff >= 0) *(double *)(ldsb + (myColBase + sCur + (int)CUR[0].dstOff)) = res;
          }
        }
        if (hdr & (1 << 6)) med_wave_sync();   // the next round reads what other lanes of this wave just wrote
      };
      {
        const int n0 = desc[0] & 15;
#define CALL(N) med_fetch<N>(desc, grec, it, ot, q, RA)
        MED_SWITCH(n0, CALL)
#undef CALL
      }
      for (int ch = 0; ch < nChunks; ch += 2) {
        body(ch, RA, RB);
        if (ch + 1 >= nChunks) break;
        body(ch + 1, RB, RA);
      }
    }
    med_wave_sync();
    // ---- halo for the next step into LDS first (its load is older than the stores below: no wait on stores) --------
    if (wantHalo) {
      double *hd = ring(slotCur, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = tid + k * blockDim.x;
        if (j < S) hd[j] = hv[k];
      }
      // a strip narrowed for short input sequences (medium_pick_geometry) may leave the workgroup with fewer than S / 4
      // threads: the rest of the supercell is copied directly
      const double *hs = A.materialise ? cellPtr(i0 - 1, t + 1) : haloIn + (long long)(t + 1) * S;
      for (int j = tid + 4 * (int)blockDim.x; j < S; j += blockDim.x) hd[j] = hs[j];
    }
    // ---- copy the finished supercell out ----------------------------------------------------------------------
    const double *cur = (const double *)(ldsb + (myColBase + sCur));
    if (active) {
      if (A.materialise) {
        double *dstp = cellPtr(i, o);
        for (int j = q; j < S; j += LPG) dstp[j] = cur[j];
      } else if (c == C - 1) {
        double *dstp = haloOut + (long long)o * S;
        for (int j = q; j < S; j += LPG) dstp[j] = cur[j];
      }
      if (i == inLen && o == outLen && q == 0 && A.loglike) A.loglike[pairIdx] = cur[P.endNode];
    }
    med_block_sync();
    slotCur = (slotCur + 1) % NS;
  }
}

// ------------------------------------------------------------------------------------------------------------
// host side: program compiler
// ------------------------------------------------------------------------------------------------------------
struct Cand { int src; int wref; };   // src: index into the LDS state vector; wref: see MedProgram::wref

std::vector<int> g_medium_cuts;
void medium_set_cuts(const std::vector<int> &cuts) { g_medium_cuts = cuts; }

static int env_int_m(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}

static double host_lse(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double mx = a > b ? a : b, mn = a > b ? b : a;
  return mx + log1p(exp(mn - mx));
}

// closure = 0: levelled (exact) program; K >= 1: silent closure in K stages -- the silent levels 1..nLev-1 are cut into K
// consecutive groups, every group is closed transitively over (a) the states finalised by earlier groups and (b) the
// emit-only parts of its own emit-fed states, and costs ONE synchronisation point.  K = 1 is the full closure.
static void build_program(const mb_machine *m, bool backward, int closure, int G, MedProgram &P, bool allowSplit = true, bool countFuse = false) {
  const int S = m->S, LPG = 64 / G, nIn = m->nIn, nOut = m->nOut;
  P = MedProgram();
  P.G = G; P.LPG = LPG; P.backward = backward; P.closure = closure != 0;
  P.NS = m->hasMatch ? 3 : 2;
  const std::vector<int> &lev = backward ? m->levB : m->levF;
  const std::vector<int> &off = backward ? m->outOff : m->inOff;
  const std::vector<uint32_t> &perm = backward ? m->outPerm : m->inPerm;
  const int nLev = backward ? m->nLevB : m->nLevF;
  const int startNode = backward ? S - 1 : 0;
  auto other = [&](uint32_t e) { return (int)(backward ? m->dst[e] : m->src[e]); };
  auto row = [&](int st, int it, int ot) { return ((long long)st * (nIn + 1) + it) * (nOut + 1) + ot; };
  const int ntok[3] = {(nIn + 1) * (nOut + 1), nIn + 1, nOut + 1};
  // emitting candidates of (node, table, token) in the reference's iteration order
  auto emitCands = [&](int st, int T, int tok, std::vector<Cand> &out) {
    out.clear();
    int it = 0, ot = 0;
    if (T == 0) { it = tok / (nOut + 1); ot = tok % (nOut + 1); if (!it || !ot) return; }
    else if (T == 1) { it = tok; if (!it) return; }
    else { ot = tok; if (!ot) return; }
    const long long rw = row(st, it, ot);
    for (int a = off[rw]; a < off[rw + 1]; ++a) out.push_back({other(perm[a]), (int)perm[a]});
  };
  // silent predecessors (a silent self-loop on state 0 never contributes to a fill, see mb_machine.cpp)
  P.silPred.assign(S, {});
  for (int st = 0; st < S; ++st) {
    const long long rw = row(st, 0, 0);
    for (int a = off[rw]; a < off[rw + 1]; ++a) {
      const int o = other(perm[a]);
      if (backward ? o <= st : o >= st) continue;
      P.silPred[st].push_back({o, perm[a]});
    }
  }
  std::vector<std::array<int, 3>> emitDeg(S);
  std::vector<Cand> tmp;
  for (int s = 0; s < S; ++s)
    for (int T = 0; T < 3; ++T) {
      int mx = 0;
      for (int tok = 0; tok < ntok[T]; ++tok) { emitCands(s, T, tok, tmp); mx = std::max(mx, (int)tmp.size()); }
      emitDeg[s][T] = mx;
    }

  // ---- nodes of the program: (destination index in the LDS vector, emit candidates of which state, "cur" candidates)
  struct Node { int dst; int emitOf; std::vector<Cand> cur; int stage; int lo = 0, hi = 1 << 30; bool seed = false; };   // [lo,hi): window of the emit candidate lists
  std::vector<Node> nodes;
  int nExtra = 0, nClosureStages = 1;
  if (!closure) {
    for (int s = 0; s < S; ++s) {
      Node n{s, s, {}, lev[s]};
      for (auto &pe : P.silPred[s]) n.cur.push_back({pe.first, (int)pe.second});
      nodes.push_back(n);
    }
  } else {
    // base states: fed by emitting edges, or the start node (which carries the seed)
    P.isBase.assign(S, 0);
    for (int s = 0; s < S; ++s) P.isBase[s] = (emitDeg[s][0] || emitDeg[s][1] || emitDeg[s][2] || s == startNode) ? 1 : 0;
    // stage of a state: 0 = no silent predecessor (its value is its emit part), else the group of its silent level
    int K = std::max(1, std::min(closure, std::max(1, nLev - 1)));
    P.stageOf.assign(S, 0);
    for (int s = 0; s < S; ++s)
      if (lev[s] > 0) P.stageOf[s] = 1 + (int)(((long long)(lev[s] - 1) * K) / std::max(1, nLev - 1));
    // explicit stage boundaries (first silent level of stages 2, 3, ...): set by the chooser (medium_set_cuts) or
    // MB_MEDIUM_CUTS="l1,l2,..." for experiments; levels need not be cut evenly -- a cut belongs where the closure is cheap
    std::vector<int> cuts = g_medium_cuts;
    if (const char *e = opt_env("MB_MEDIUM_CUTS")) { cuts.clear(); for (const char *q = e; *q;) { cuts.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
    if (!cuts.empty()) {
      std::sort(cuts.begin(), cuts.end());
      for (int s = 0; s < S; ++s)
        if (lev[s] > 0) { int st = 1; for (int c : cuts) if (lev[s] >= c) ++st; P.stageOf[s] = st; }
      K = (int)cuts.size() + 1;
    }
    const std::vector<int> &stg = P.stageOf;
    // closure structure: ancestors through silent paths whose intermediate states lie in the node's own stage, in
    // topological order.  An ancestor is either final already (earlier stage) or an emit-fed state of the same stage
    // (then its emit-only part is the source).
    P.closBase.assign(S, {}); P.closPair.assign(S, {});
    std::vector<int> order(S);
    std::iota(order.begin(), order.end(), 0);
    if (backward) std::reverse(order.begin(), order.end());
    for (int d : order) {
      std::vector<int> anc;
      for (auto &pe : P.silPred[d]) {
        const int sp = pe.first;
        if (stg[sp] < stg[d]) { anc.push_back(sp); continue; }
        if (P.isBase[sp]) anc.push_back(sp);
        anc.insert(anc.end(), P.closBase[sp].begin(), P.closBase[sp].end());
      }
      std::sort(anc.begin(), anc.end());
      anc.erase(std::unique(anc.begin(), anc.end()), anc.end());
      P.closBase[d] = anc;
      for (size_t k = 0; k < anc.size(); ++k) P.closPair[d].push_back(P.nPairs++);
    }
    // e-slots: a base state that also has silent predecessors keeps its emit-only part in an extra vector entry
    std::vector<int> eslot(S, -1);
    for (int s = 0; s < S; ++s)
      if (P.isBase[s] && !P.silPred[s].empty()) eslot[s] = S + 1 + nExtra++;
    auto srcIdx = [&](int b, int d) { return (stg[b] < stg[d] || eslot[b] < 0) ? b : eslot[b]; };
    for (int s = 0; s < S; ++s) {
      if (P.isBase[s] || P.silPred[s].empty()) nodes.push_back(Node{eslot[s] >= 0 ? eslot[s] : s, s, {}, 0});   // stage 0 (also dead states)
      if (!P.silPred[s].empty()) {
        Node n{s, -1, {}, stg[s]};
        if (P.isBase[s]) n.cur.push_back({eslot[s], -2 - P.nPairs});   // own emit part, weight 0 (a constant "pair")
        for (size_t k = 0; k < P.closBase[s].size(); ++k) n.cur.push_back({srcIdx(P.closBase[s][k], s), -2 - P.closPair[s][k]});
        nodes.push_back(n);
      }
    }
    nClosureStages = K;
  }
  // ---- split high-degree nodes -------------------------------------------------------------------------------------------
  // A state with dozens of emitting candidates for one token (psw2dna, Backward: three states fan out to 62 codon
  // states) would make its whole round 60+ slots deep while one lane works.  Its candidate lists are cut into parts of
  // `part`, each evaluated by its own lane into an extra vector entry (stage 0: emitting candidates read other cells
  // only), and a combining node folds the partial results with weight 0 one stage later -- max of maxes and
  // log-sum of log-sums are the same reductions, and for max the result is bit-identical.
  bool anySplit = false, anySplitCur = false;
  int extraStages = 0;
  {
    const int splitAt = allowSplit ? env_int_m("MB_MEDIUM_SPLIT_DEGREE", 8) : 0, part = std::max(2, env_int_m("MB_MEDIUM_SPLIT_PART", 8));
    const int seedStage = closure ? 0 : lev[startNode];
    std::vector<Node> outNodes;
    for (Node &n : nodes) {
      if (n.emitOf == startNode && n.stage == seedStage) n.seed = true;
      const int md = n.emitOf >= 0 ? std::max(emitDeg[n.emitOf][0], std::max(emitDeg[n.emitOf][1], emitDeg[n.emitOf][2])) : 0;
      if (splitAt <= 0 || md <= splitAt) { outNodes.push_back(n); continue; }
      anySplit = true;
      Node comb = n;
      comb.emitOf = -1; comb.stage = -1 - n.stage;      // marked: becomes n.stage + 1 below (and is never left in stage 0)
      for (int lo = 0; lo < md; lo += part) {
        Node pn{S + 1 + nExtra, n.emitOf, {}, 0};
        pn.lo = lo; pn.hi = lo + part;
        comb.cur.push_back({S + 1 + nExtra, -2 - P.nPairs});   // the constant-0 "pair"
        ++nExtra;
        outNodes.push_back(pn);
      }
      outNodes.push_back(comb);
    }
    if (anySplit) {
      for (Node &n : outNodes) {
        if (n.stage < 0) n.stage = -n.stage;            // combining node: old stage + 1
        else if (n.stage > 0) n.stage += 1;             // everything that may depend on a combined value moves one stage on
      }
      nodes.swap(outNodes);
    }
    // the same for long lists of same-cell candidates (silent fan-in): the parts stay in the node's stage (they read
    // values of earlier stages), the combining node opens a new stage right behind it
    if (splitAt > 0) {
      int maxStage = 0;
      for (const Node &n : nodes) maxStage = std::max(maxStage, n.stage);
      std::vector<int> shiftOf(maxStage + 2, 0);
      for (int st = 0, shift = 0; st <= maxStage; ++st) {
        shiftOf[st] = shift;
        bool need = false;
        for (const Node &n : nodes) if (n.stage == st && (int)n.cur.size() > splitAt) need = true;
        if (need) ++shift;
        shiftOf[st + 1] = shift;
      }
      std::vector<Node> out2;
      for (Node &n : nodes) {
        const int st = n.stage;
        if ((int)n.cur.size() <= splitAt) { n.stage = st + shiftOf[st]; out2.push_back(n); continue; }
        Node comb = n;
        comb.cur.clear(); comb.stage = st + shiftOf[st] + 1;
        for (size_t lo = 0; lo < n.cur.size(); lo += part) {
          Node pn{S + 1 + nExtra, -1, {}, st + shiftOf[st]};
          pn.cur.assign(n.cur.begin() + lo, n.cur.begin() + std::min(n.cur.size(), lo + part));
          comb.cur.push_back({S + 1 + nExtra, -2 - P.nPairs});
          ++nExtra;
          out2.push_back(pn);
        }
        out2.push_back(comb);
        anySplitCur = true;
      }
      if (anySplitCur) { nodes.swap(out2); extraStages = shiftOf[maxStage + 1]; }
    }
  }
  // [S] = -inf sentinel, the e-slots, one dummy entry idle lanes write to.  Even length keeps every column 16-byte
  // aligned; with one or two lanes per supercell the lanes of a wavefront read DIFFERENT columns at the same state
  // offset, and an odd length (stride of 2 x odd LDS banks) makes those reads conflict-free.
  P.hasSplits = anySplit || anySplitCur;
  // count programs (flat form): the usage of the emitting transitions can ride on the fill's emit rounds when every emitting candidate
  // sits in the node of its real destination state (no parts of split candidate lists) -- decided here, finished below once the rounds exist
  bool fuse = countFuse && !backward && !anySplit;
  P.Spad = (S + 1 + nExtra + 1 + 1) & ~1;
  if (LPG <= 2) P.Spad |= 1;
  P.dummyOff = (uint32_t)(S + 1 + nExtra) * 8u;
  const int nStages = (closure ? 1 + nClosureStages : nLev) + (anySplit ? 1 : 0) + extraStages;

  // ---- rounds: per stage, nodes sorted so that a round is homogeneous in (tables used, candidate count) ----------
  auto sig = [&](const Node &n) {
    std::array<int, 4> d{0, 0, 0, (int)n.cur.size()};
    if (n.emitOf >= 0) for (int T = 0; T < 3; ++T) d[T] = std::max(0, std::min(emitDeg[n.emitOf][T], n.hi) - n.lo);
    return d;
  };
  std::vector<std::vector<int>> rounds;
  std::vector<unsigned char> sync;
  for (int st = 0; st < nStages; ++st) {
    std::vector<int> ids;
    for (int k = 0; k < (int)nodes.size(); ++k) if (nodes[k].stage == st) ids.push_back(k);
    // order by candidate signature, then cut the ordered list into rounds of <= LPG nodes by dynamic programming:
    // a round costs a fixed overhead plus, per table, as many slots as its most demanding node (padding is work).
    std::stable_sort(ids.begin(), ids.end(), [&](int x, int y) { return sig(nodes[x]) > sig(nodes[y]); });
    const int nIds = (int)ids.size();
    const double roundCost = 6.0, slotCost = 9.0, singleCost = 3.0;
    std::vector<double> best(nIds + 1, 1e300);
    std::vector<int> cut(nIds + 1, 0);
    best[0] = 0.0;
    for (int e = 1; e <= nIds; ++e) {
      std::array<int, 4> mxs{0, 0, 0, 0};
      for (int b = e - 1; b >= 0 && e - b <= LPG; --b) {
        const auto d = sig(nodes[ids[b]]);
        for (int T = 0; T < 4; ++T) mxs[T] = std::max(mxs[T], d[T]);
        const int tot = mxs[0] + mxs[1] + mxs[2] + mxs[3];
        const double c = best[b] + roundCost + (tot <= 1 ? singleCost : tot * slotCost);
        if (c < best[e]) { best[e] = c; cut[e] = b; }
      }
    }
    std::vector<std::pair<int, int>> segs;
    for (int e = nIds; e > 0; e = cut[e]) segs.push_back({cut[e], e});
    std::reverse(segs.begin(), segs.end());
    for (auto &sg : segs) {
      rounds.emplace_back(ids.begin() + sg.first, ids.begin() + sg.second);
      sync.push_back(0);
    }
    if (!sync.empty()) sync.back() = 1;
  }
  if (!sync.empty()) sync.back() = 0;
  P.nRounds = (int)rounds.size();

  // ---- descriptors and records -------------------------------------------------------------------------------------
  const int mulI[4] = {LPG * (nOut + 1), LPG, 0, 0}, mulO[4] = {LPG, 0, LPG, 0};
  const int ntokT[4] = {ntok[0], ntok[1], ntok[2], 1};
  const MedRec padRec{-INFINITY, (uint32_t)S * 8u, (uint32_t)(S + 1 + nExtra) * 8u};   // idle lanes store into the dummy entry
  auto candsOf = [&](const Node &n, int T, int tok, std::vector<Cand> &out) {
    if (T == 3) { out = n.cur; return; }
    out.clear();
    if (n.emitOf >= 0) {
      emitCands(n.emitOf, T, tok, out);
      if (n.lo > 0 || n.hi < (int)out.size()) {      // this node's window of the candidate list
        const int a = std::min<int>(n.lo, (int)out.size()), b = std::min<int>(n.hi, (int)out.size());
        out = std::vector<Cand>(out.begin() + a, out.begin() + b);
      }
    }
  };
  for (int r = 0; r < P.nRounds; ++r) {
    // slots of the round, table by table; a chunk holds up to MS slots of ONE table (so it reads one LDS vector)
    int nsT[4], total = 0;
    for (int T = 0; T < 4; ++T) {
      nsT[T] = 0;
      for (int id : rounds[r]) nsT[T] = std::max(nsT[T], sig(nodes[id])[T]);
      total += nsT[T];
    }
    if (env_int_m("MB_MEDIUM_JIT_VERBOSE", 0) >= 2) {
      int used = 0;
      for (int id : rounds[r]) { const auto d = sig(nodes[id]); used += d[0] + d[1] + d[2] + d[3]; }
      fprintf(stderr, "[mbhip]   round %d: stage %d, %zu of %d lanes, slots match/in/out/cur %d/%d/%d/%d, %d of %d lane-slots carry a candidate\n", r, nodes[rounds[r][0]].stage,
              rounds[r].size(), LPG, nsT[0], nsT[1], nsT[2], nsT[3], used, total * LPG);
    }
    bool single = (total == 1);
    for (int id : rounds[r]) { const auto d = sig(nodes[id]); if (d[0] + d[1] + d[2] + d[3] > 1) single = false; }
    if (total == 0) { nsT[3] = 1; total = 1; single = true; }   // a round of dead states still writes -inf through one padded slot
    P.roundInfo.emplace_back();
    P.roundInfo.back().sync = sync[r] != 0;
    P.roundInfo.back().single = single;
    P.roundInfo.back().fused = fuse && (nsT[0] + nsT[1] + nsT[2] > 0);
    if (P.roundInfo.back().fused && total > medium_jit_max_cands()) fuse = false;      // (the generator's running-form rounds carry no usage terms: decided for the whole program below)
    struct Ch { int T, j0, n; };
    std::vector<Ch> chs;
    for (int T = 0; T < 4; ++T)
      for (int j0 = 0; j0 < nsT[T]; j0 += MS) chs.push_back({T, j0, std::min(MS, nsT[T] - j0)});
    for (size_t ci = 0; ci < chs.size(); ++ci) {
      const Ch &c = chs[ci];
      const bool firstC = ci == 0, lastC = ci + 1 == chs.size();
      const size_t db = P.desc.size();
      P.desc.resize(db + DW, 0);
      P.desc[db] = c.n | (firstC << 4) | (lastC << 5) | ((lastC && sync[r]) << 6) | (15 << 8) | (single << 12);
      if (db) P.desc[db - DW] = (P.desc[db - DW] & ~(15 << 8)) | (c.n << 8);   // nsNext of the previous chunk
      const size_t b0 = P.rec.size();
      const int stride = ntokT[c.T] * LPG;
      P.desc[db + 1] = (int)b0;
      P.desc[db + 2] = mulI[c.T];
      P.desc[db + 3] = mulO[c.T] | (c.T << 24);
      P.desc[db + 4] = stride;
      P.rec.resize(b0 + (size_t)c.n * stride, padRec);
      P.wref.resize(b0 + (size_t)c.n * stride, -1);
      P.recT.resize(b0 + (size_t)c.n * stride, (unsigned char)c.T);
      for (int k = 0; k < c.n; ++k) {
        P.roundInfo.back().slots.push_back({c.T, (long long)b0 + (long long)k * stride});
      }
      for (size_t ln = 0; ln < rounds[r].size(); ++ln) {
        const Node &n = nodes[rounds[r][ln]];
        for (int tok = 0; tok < ntokT[c.T]; ++tok) {
          candsOf(n, c.T, tok, tmp);
          for (int k = 0; k < c.n; ++k) {
            const size_t idx = b0 + (size_t)k * stride + (size_t)tok * LPG + ln;
            if (k == 0) P.rec[idx].dstOff = (uint32_t)n.dst * 8u | ((countFuse && n.emitOf >= 0) ? ((uint32_t)n.emitOf * 8u) << 16 : 0u);      // (upper half: the real state's place in the Backward supercell)
            const int j = c.j0 + k;
            if (j < (int)tmp.size()) {
              P.rec[idx].srcOff = (uint32_t)tmp[j].src * 8u; P.wref[idx] = tmp[j].wref;
              if (c.T < 2) P.haloStates.push_back(tmp[j].src);      // read from the column to the left: crosses a strip boundary
            }
          }
        }
      }
    }
  }
  std::sort(P.haloStates.begin(), P.haloStates.end());
  P.haloStates.erase(std::unique(P.haloStates.begin(), P.haloStates.end()), P.haloStates.end());
  {   // in-place ring (mb_medium.h): the emit rounds must all lie in the first stage, in the straight-line form, and be few
    int firstSync = -1, lastEmit = -1; long long slots0 = 0; bool anyBig = false;
    for (size_t r = 0; r < P.roundInfo.size(); ++r) {
      const MedRoundInfo &ri = P.roundInfo[r];
      bool emit = false;
      for (const MedSlotInfo &sl : ri.slots) emit = emit || sl.T < 3;
      if (emit) { lastEmit = (int)r; if ((int)ri.slots.size() > medium_jit_max_cands()) anyBig = true; }
      if (firstSync < 0) { slots0 += (long long)ri.slots.size(); if (ri.sync) firstSync = (int)r; }
    }
    if (firstSync < 0) firstSync = (int)P.roundInfo.size() - 1;
    P.inPlaceOk = closure != 0 && lastEmit >= 0 && lastEmit <= firstSync && !anyBig && slots0 <= env_int_m("MB_MEDIUM_INPLACE_MAXSLOTS", 48);
  }
  P.fusedEmit = fuse;
  if (!fuse) for (MedRoundInfo &ri : P.roundInfo) ri.fused = false;
  if (countFuse && !fuse) for (MedRec &r : P.rec) r.dstOff &= 0xFFFFu;
  P.nChunks = (int)(P.desc.size() / DW);
  // where the seed goes: the stage-1 destination of the start node
  P.dev.seedOff = 0;
  for (const Node &n : nodes)
    if (n.seed) P.dev.seedOff = (unsigned)n.dst * 8u;
}

template <class T>
static bool up(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D(program)")) return false;
  return true;
}

// (Re)evaluate every record's log-weight: plain edges take logW[edge]; closure pairs take the log-sum over all silent
// paths from the base state to the node (recurrence over the silent DAG in topological order).  Host only.
void medium_eval_weights(const mb_machine *m, MedProgram &P) {
  std::vector<double> pairW(P.nPairs + 1, -INFINITY);
  pairW[P.nPairs] = 0.0;   // the constant "own emit part" pair
  if (P.closure) {
    const int S = m->S;
    std::vector<int> order(S);
    std::iota(order.begin(), order.end(), 0);
    if (P.backward) std::reverse(order.begin(), order.end());
    for (int d : order) {
      const std::vector<int> &anc = P.closBase[d];
      for (auto &pe : P.silPred[d]) {
        const int sp = pe.first;
        const double w = m->logW[pe.second];
        auto addTo = [&](int b, double x) {
          const size_t k = std::lower_bound(anc.begin(), anc.end(), b) - anc.begin();
          double &t = pairW[P.closPair[d][k]];
          t = host_lse(t, x);
        };
        if (P.stageOf[sp] < P.stageOf[d]) { addTo(sp, w); continue; }   // final already: paths through it belong to its own value
        if (P.isBase[sp]) addTo(sp, w);
        for (size_t k = 0; k < P.closBase[sp].size(); ++k) addTo(P.closBase[sp][k], pairW[P.closPair[sp][k]] + w);
      }
    }
  }
  for (size_t k = 0; k < P.rec.size(); ++k) {
    const int r = P.wref[k];
    P.rec[k].w = r >= 0 ? m->logW[r] : (r == -1 ? -INFINITY : pairW[-2 - r]);
  }
  // the in-place ring's copy (MedProgram::recC): sources of the input-consuming tables by their place in the short vector (padding: the
  // short vector's own -inf entry), the destination's place + 1 in the upper half of dstOff
  P.recC.clear();
  if (!P.counting && P.inPlaceOk && P.Spad * 8 < (1 << 16) && P.recT.size() == P.rec.size()) {
    std::vector<int> place((size_t)P.Spad + 2, -1);
    for (size_t k = 0; k < P.haloStates.size(); ++k) place[(size_t)P.haloStates[k]] = (int)k;
    const uint32_t KH = (uint32_t)P.haloStates.size();
    P.recC = P.rec;
    for (size_t k = 0; k < P.recC.size(); ++k) {
      MedRec &r = P.recC[k];
      if (P.recT[k] < 2) { const uint32_t st = r.srcOff >> 3; r.srcOff = ((st < place.size() && place[st] >= 0) ? (uint32_t)place[st] : KH) * 8u; }
      const uint32_t d = r.dstOff >> 3;
      if (d < place.size() && place[d] >= 0) r.dstOff |= ((uint32_t)(place[d] + 1) * 8u) << 16;
    }
  }
}

bool medium_refresh_weights(const mb_machine *m, MedProgram &P) {
  medium_eval_weights(m, P);
  if (!up(P.d_rec, P.rec)) return false;
  P.dev.rec = P.d_rec;
  std::vector<MedRec> img(P.ldsImageIdx.size());
  for (size_t k = 0; k < img.size(); ++k) img[k] = P.rec[P.ldsImageIdx[k]];
  if (!up(P.d_ldsImage, img)) return false;
  P.dev.ldsImage = P.d_ldsImage;
  if (P.counting) { if (!up(P.d_accMap, P.accMap)) return false; P.dev.accMap = P.d_accMap; }
  if (!P.recC.empty()) {
    if (!up(P.d_recC, P.recC)) return false;
    std::vector<MedRec> imgC(P.ldsImageIdx.size());
    for (size_t k = 0; k < imgC.size(); ++k) imgC[k] = P.recC[P.ldsImageIdx[k]];
    if (!up(P.d_ldsImageC, imgC)) return false;
  }
  return true;
}

bool medium_build_host(const mb_machine *m, bool backward, int closure, int G, MedProgram &P, MedGeom &geo) {
  build_program(m, backward, closure, G, P);
  if (P.rec.size() >= (1u << 30) || P.Spad * 8 >= (1 << 24)) { set_error("machine too large for the tiled kernel family"); return false; }
  MedProgDev &d = P.dev;
  d.S = m->S; d.Spad = P.Spad; d.LPG = P.LPG; d.G = G; d.NS = P.NS; d.nChunks = P.nChunks;
  d.nIn = m->nIn; d.nOut = m->nOut;
  d.startNode = backward ? m->S - 1 : 0; d.endNode = backward ? 0 : m->S - 1;
  if (!medium_geometry(m, P, geo)) { set_error("machine does not fit the tiled kernel family's LDS ring"); return false; }
  medium_fit_records(m, P, geo);
  medium_jit_plan(m, P, geo);
  medium_eval_weights(m, P);
  return true;
}

// The geometry above takes as many columns as the LDS ring allows.  On machines with many candidate slots that leaves no
// LDS for the candidate records, which then stream from L2 on every step (protpsw.translate.dnapsw: 42 KB per wave-step,
// 670 KB per step and CU -- the L1 fill rate, not HBM, bounded the kernel).  Give columns up -- down to half -- until
// the records that cannot sit in VGPRs fit next to the ring.
void medium_fit_records(const mb_machine *m, const MedProgram &P, MedGeom &geo) {
  const char *e = opt_env("MB_MEDIUM_MAXWAVES");
  if (e && atoi(e) > 0 && atoi(e) < geo.waves) { geo.waves = atoi(e); geo.C = geo.waves * P.G; }
  if (opt_env("MB_MEDIUM_FIT_RECORDS") && atoi(opt_env("MB_MEDIUM_FIT_RECORDS")) == 0) return;
  if (P.counting && env_int_m("MB_MEDIUM_COUNT_FIT", 1) == 0) return;
  const long long ntokT[4] = {(long long)(m->nIn + 1) * (m->nOut + 1), m->nIn + 1, m->nOut + 1, 1};
  long long slotsT[4] = {0, 0, 0, 0};
  for (const MedRoundInfo &ri : P.roundInfo) for (const MedSlotInfo &sl : ri.slots) ++slotsT[sl.T];
  const int minWaves = std::max(1, (geo.waves + 1) / 2);
  // machines with a handful of states: a step is shorter than the acknowledgement of the previous step's stores, and on
  // gfx9 any vector load waits for those (one in-order vmcnt) -- so the halo supercells of a whole tile are fetched in
  // the tile's prologue instead (MedGeom::haloSteps), if 32 KB of LDS buy that
  const bool wantHaloTile = env_int_m("MB_MEDIUM_HALO_TILE", 1) != 0 && !P.counting;
  auto haloStepsFor = [&](int C) { const long long cap = std::max(C, 128); return (wantHaloTile && cap * m->S * 8 <= 32 * 1024) ? (int)cap : 0; };
  while (true) {
    const int regBudget = std::max(0, std::min(512 / ((geo.waves + 3) / 4), 256) - 84);
    const long long perRec = P.counting ? 5 : 4;
    const long long inReg = std::max<long long>(0, regBudget / perRec - slotsT[1]);      // token-independent records the VGPRs can hold
    const long long need = (slotsT[2] * ntokT[2] + std::max<long long>(0, slotsT[3] - inReg)) * P.LPG * 16 + (long long)haloStepsFor(geo.C) * m->S * 8;
    const long long ring = (long long)(P.NS + (P.counting ? 1 : 0)) * (geo.C + 1) * P.Spad * 8 + (P.counting ? (long long)(P.accEntries + 2) * 8 : 0);
    if (ring + need + 2048 <= 160 * 1024 || geo.waves <= minWaves) break;
    --geo.waves; geo.C = geo.waves * P.G;
  }
  geo.haloSteps = haloStepsFor(geo.C);
  if (geo.haloSteps && (long long)P.NS * (geo.C + 1) * P.Spad * 8 + (long long)geo.haloSteps * m->S * 8 + 4096 > 160 * 1024) geo.haloSteps = 0;
  geo.ldsBytes = (size_t)P.NS * (geo.C + 1) * P.Spad * sizeof(double);
}

// the exact Forward program with every state's candidates in ONE node (no parts / combining nodes): what the traceback-byte
// Viterbi sweep needs when the ordinary exact program splits high-degree states (a byte names a candidate of the state itself)
bool medium_build_unsplit(const mb_machine *m, int G, MedProgram &P, MedGeom &geo) {
  build_program(m, false, 0, G, P, /*allowSplit=*/false);
  if (P.rec.size() >= (1u << 30) || P.Spad * 8 >= (1 << 24)) return false;
  MedProgDev &d = P.dev;
  d.S = m->S; d.Spad = P.Spad; d.LPG = P.LPG; d.G = G; d.NS = P.NS; d.nChunks = P.nChunks;
  d.nIn = m->nIn; d.nOut = m->nOut; d.startNode = 0; d.endNode = m->S - 1;
  if (!medium_geometry(m, P, geo)) return false;
  medium_fit_records(m, P, geo);
  medium_jit_plan(m, P, geo);
  medium_eval_weights(m, P);
  if (!up(P.d_desc, P.desc)) return false;
  P.dev.desc = P.d_desc;
  return medium_refresh_weights(m, P);
}

bool medium_build(const mb_machine *m, bool backward, int closure, int G, MedProgram &P, MedGeom &geo) {
  if (!medium_build_host(m, backward, closure, G, P, geo)) return false;
  if (!up(P.d_desc, P.desc)) return false;
  P.dev.desc = P.d_desc;
  return medium_refresh_weights(m, P);
}

// Count programs.  MachineCounts (src/counts.cpp:57-64, src/backward.cpp:58-87) adds exp(F(src cell, src) + w + B(dst cell, dst) - LL)
// per transition and cell.  Two forms:
//   LEVELLED (round 2/3, MB_MEDIUM_COUNT_FLAT=0): the exact Forward program, every candidate of its log-sum-exp is also a usage
//     term of its transition; the accumulator offset rides in the upper half of srcOff.  The program is as deep as the machine's
//     silent levels and its rounds are as wide as a level: psw2dna runs 17 rounds / 27 candidate slots per supercell with half
//     of the lanes idle, although only ~276 of its 1684 transitions apply to a given cell (in-degree 1 for 254 of 271 states).
//   FLAT (default): the fill is the staged-CLOSURE Forward program (the rounds of the log-likelihood sweep: 2 synchronisation
//     points instead of 10 for psw2dna), and the usage terms follow in ONE more round without any dependency between lanes:
//     the transitions that apply to a cell -- by table: match (input, output token), input-token, output-token, silent -- are
//     dealt to the lanes of the column, ceil(n / LPG) slots per table, every record naming its own source (in the vector its
//     table reads), destination (in the Backward supercell) and accumulator.  F(src) of EVERY state is final in the ring by
//     then (closure programs finalise all S states), so the term is the reference's, with the Forward sweep's own rounding.
static void append_flat_usage(const mb_machine *m, MedProgram &P) {
  const int LPG = P.LPG, nIn = m->nIn, nOut = m->nOut, S = m->S;
  const long long ntokT[4] = {(long long)(nIn + 1) * (nOut + 1), nIn + 1, nOut + 1, 1};
  std::vector<std::vector<uint32_t>> byTok[4];
  for (int T = 0; T < 4; ++T) byTok[T].resize((size_t)ntokT[T]);
  for (long long e = 0; e < m->nTrans; ++e) {
    const int it = m->inTok[e], ot = m->outTok[e];
    const int T = (it && ot) ? 0 : (it ? 1 : (ot ? 2 : 3));
    if (T == 3 && m->dst[e] <= m->src[e]) continue;   // the silent self-loop on state 0: no candidate of any fill (mb_machine.cpp)
    const long long tok = T == 0 ? (long long)it * (nOut + 1) + ot : (T == 1 ? it : (T == 2 ? ot : 0));
    byTok[T][(size_t)tok].push_back((uint32_t)e);
  }
  P.roundInfo.emplace_back();
  MedRoundInfo &ri = P.roundInfo.back();
  ri.flat = true;
  for (int T = 0; T < 4; ++T) {
    if (P.fusedEmit && T < 3) continue;      // their usage terms ride on the fill's emit rounds
    size_t mx = 0;
    for (auto &l : byTok[T]) {
      // lanes side by side read B(dst) (and mostly F(src)) of neighbouring states: no LDS bank is asked twice
      std::stable_sort(l.begin(), l.end(), [&](uint32_t a, uint32_t b) { return m->dst[a] < m->dst[b]; });
      mx = std::max(mx, l.size());
    }
    const int ns = (int)((mx + LPG - 1) / LPG);
    for (int k = 0; k < ns; ++k) {
      const size_t b0 = P.rec.size();
      P.rec.resize(b0 + (size_t)ntokT[T] * LPG);
      P.wref.resize(b0 + (size_t)ntokT[T] * LPG, -1);
      P.recT.resize(b0 + (size_t)ntokT[T] * LPG, (unsigned char)T);
      ri.slots.push_back({T, (long long)b0});
      for (long long tok = 0; tok < ntokT[T]; ++tok)
        for (int ln = 0; ln < LPG; ++ln) {
          MedRec &r = P.rec[b0 + (size_t)tok * LPG + ln];
          const size_t j = (size_t)k * LPG + ln;
          const std::vector<uint32_t> &l = byTok[T][(size_t)tok];
          if (j < l.size()) {
            const uint32_t e = l[j];
            r.w = m->logW[e]; r.srcOff = (uint32_t)m->src[e] * 8u | ((uint32_t)m->dst[e] * 8u) << 16; r.dstOff = e * 8u;
            P.wref[b0 + (size_t)tok * LPG + ln] = (int)e;
          } else {   // padding: exp(-inf) = 0 into the lane's own dummy accumulator
            r.w = -INFINITY; r.srcOff = (uint32_t)S * 8u | (P.dummyOff << 16); r.dstOff = (uint32_t)(m->nTrans + ln) * 8u;
          }
        }
    }
  }
  if (ri.slots.empty()) P.roundInfo.pop_back();
}

// Geometry leaves room for one Backward supercell per column and the count array in LDS, and keeps the workgroup at 8 wavefronts.
bool medium_build_count_host(const mb_machine *m, int G, MedProgram &P, MedGeom &geo, int closure, const std::vector<int> &cuts) {
  const bool flat = env_int_m("MB_MEDIUM_COUNT_FLAT", 1) != 0;
  if (flat) {
    medium_set_cuts(cuts);
    build_program(m, false, closure, G, P, /*allowSplit=*/true, /*countFuse=*/env_int_m("MB_MEDIUM_COUNT_FUSE", 1) != 0);
    medium_set_cuts({});
  } else
    build_program(m, false, 0, G, P, /*allowSplit=*/false);   // counting terms need every candidate beside its real destination
  P.counting = true; P.flatCount = flat; P.accEntries = (int)m->nTrans + P.LPG;   // + one dummy accumulator per lane of a group (padding candidates)
  if (P.rec.size() >= (1u << 30) || P.Spad * 8 >= (1 << 16) || (m->nTrans + 64 + 2) * 8 >= (1 << 16)) return false;
  MedProgDev &d = P.dev;
  d.S = m->S; d.Spad = P.Spad; d.LPG = P.LPG; d.G = G; d.NS = P.NS; d.nChunks = P.nChunks;
  d.nIn = m->nIn; d.nOut = m->nOut; d.startNode = 0; d.endNode = m->S - 1;
  if (flat) append_flat_usage(m, P);
  if (!medium_geometry(m, P, geo)) return false;
  medium_fit_records(m, P, geo);
  if (!flat)
    for (size_t k = 0; k < P.rec.size(); ++k) {
      const long long e = P.wref[k] >= 0 ? P.wref[k] : m->nTrans + (long long)(k % P.LPG);   // padding candidates add 0 to their lane's dummy accumulator
      P.rec[k].srcOff = (P.rec[k].srcOff & 0xFFFFu) | ((uint32_t)(e * 8) << 16);
    }
  medium_jit_plan(m, P, geo);
  medium_count_layout(m, P);
  if (flat && env_int_m("MB_MEDIUM_COUNT_COMPACT", 1)) {
    // the placement says which usage records are loop-invariant: only the others need an accumulator that lives through the step loop.
    // With that (smaller) table the geometry is taken again -- more columns -- and the placement with it.
    MedGeom geo2;
    if (medium_geometry(m, P, geo2)) {
      medium_fit_records(m, P, geo2);
      geo = geo2;
      P.planC = 0; P.planHalo = 0; P.planWaves = 0; P.regBudget = -1;
      medium_jit_plan(m, P, geo);
      medium_count_layout(m, P);
    }
  }
  medium_eval_weights(m, P);
  return true;
}

// Accumulator offsets of a flat count program's usage records, by placement (after every medium_jit_plan).  A usage record held in
// VGPRs sums in a register over a tile and reaches an accumulator AFTER the step loop: entry e of the all-transition table laid over
// the dead ring.  Every other usage record adds per step (ds_add): entry compact(e) of the loop-time table, compact ids dealt in
// order of first appearance (accMap: entry -> transition).  Flat records carry the offset in dstOff, fused emit records in the
// upper half of srcOff.  MB_MEDIUM_COUNT_COMPACT=0: one table of nTrans + LPG entries for both, as in round 4.
void medium_count_layout(const mb_machine *m, MedProgram &P) {
  if (!P.counting || !P.flatCount) return;
  const bool compact = env_int_m("MB_MEDIUM_COUNT_COMPACT", 1) != 0;
  const int LPG = P.LPG;
  const long long ntokT[4] = {(long long)(m->nIn + 1) * (m->nOut + 1), m->nIn + 1, m->nOut + 1, 1};
  std::vector<int> compactOf((size_t)m->nTrans, -1);
  P.accMap.clear();
  auto isUsage = [&](const MedRoundInfo &ri, const MedSlotInfo &sl) { return ri.flat || (ri.fused && sl.T < 3); };
  if (compact)
    for (const MedRoundInfo &ri : P.roundInfo)
      for (const MedSlotInfo &sl : ri.slots) {
        if (!isUsage(ri, sl) || sl.place == MED_PLACE_REG) continue;
        for (long long k = 0; k < ntokT[sl.T] * LPG; ++k) {
          const int e = P.wref[(size_t)(sl.recBase + k)];
          if (e >= 0 && compactOf[(size_t)e] < 0) { compactOf[(size_t)e] = (int)P.accMap.size(); P.accMap.push_back(e); }
        }
      }
  else
    for (long long e = 0; e < m->nTrans; ++e) { compactOf[(size_t)e] = (int)e; P.accMap.push_back((int)e); }
  const int nLoop = (int)P.accMap.size();
  P.accEntries = nLoop + LPG;
  P.accAllEntries = compact ? (int)m->nTrans + LPG : 0;
  for (const MedRoundInfo &ri : P.roundInfo)
    for (const MedSlotInfo &sl : ri.slots) {
      if (!isUsage(ri, sl)) continue;
      const bool inReg = compact && sl.place == MED_PLACE_REG;
      for (long long k = 0; k < ntokT[sl.T] * LPG; ++k) {
        MedRec &r = P.rec[(size_t)(sl.recBase + k)];
        const int e = P.wref[(size_t)(sl.recBase + k)];
        const int ln = (int)(k % LPG);
        const uint32_t off = (uint32_t)(e >= 0 ? (inReg ? e : compactOf[(size_t)e]) : (inReg ? (int)m->nTrans : nLoop) + ln) * 8u;      // padding: the lane's dummy entry
        if (ri.flat) r.dstOff = off;
        else r.srcOff = (r.srcOff & 0xFFFFu) | (off << 16);
      }
    }
}

bool medium_build_count(const mb_machine *m, int G, MedProgram &P, MedGeom &geo, int closure, const std::vector<int> &cuts) {
  if (!medium_build_count_host(m, G, P, geo, closure, cuts)) return false;
  if (!up(P.d_desc, P.desc)) return false;
  P.dev.desc = P.d_desc;
  return medium_refresh_weights(m, P);
}

void medium_free(MedProgram &P) {
  medium_jit_free(P);
  void *ptrs[] = {P.d_desc, P.d_rec, P.d_ldsImage, P.d_accMap, P.d_recC, P.d_ldsImageC};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  P = MedProgram();
}

// Geometry: columns per strip limited by the 160 KB LDS of a CU.
bool medium_geometry(const mb_machine *m, const MedProgram &P, MedGeom &geo) {
  // count programs keep one Backward supercell per column and the count array next to the ring
  const size_t perCol = (size_t)(P.NS + (P.counting ? 1 : 0)) * P.Spad * sizeof(double);
  const size_t progBytes = 0;
  const size_t fixed = 512 + (P.counting ? (size_t)(std::max(P.accEntries, P.LPG) + 2) * sizeof(double) + 1024 : 0);
  if (fixed + 2 * perCol > 160 * 1024) return false;
  const size_t budget = 160 * 1024 - fixed;
  long long maxCols = (long long)(budget / perCol) - 1;   // one extra column for the halo
  if (maxCols < P.G) return false;
  // one or two lanes per supercell (machines with a handful of states): a 512-column strip wastes half its steps on
  // the skewed start/end of a 1 kb sweep; 8 wavefronts (256 columns) measured best on dnapsw / protpsw
  int waves = (int)std::min<long long>(maxCols / P.G, P.counting ? std::max(1, std::min(16, env_int_m("MB_MEDIUM_COUNT_MAXWAVES", 8))) : (P.LPG <= 2 ? 8 : 16));
  // S must be covered by 4 halo registers per thread
  while (waves < 16 && (long long)waves * 64 * 4 < m->S && (long long)(waves + 1) * P.G <= maxCols) ++waves;
  if ((long long)waves * 64 * 4 < m->S || (long long)waves * P.G > maxCols) return false;
  geo.waves = waves; geo.C = waves * P.G;
  geo.ldsBytes = (size_t)P.NS * (geo.C + 1) * P.Spad * sizeof(double) + progBytes;
  return true;
}

static bool launch_jit(const MedJit &J, dim3 grid, dim3 block, hipStream_t st, const MedProgDev &P, const MedTileArgs &A) {
  if (!J.func) return false;
  MedProgDev p = P; MedTileArgs a = A;
  void *args[] = {&p, &a};
  return hipModuleLaunchKernel((hipFunction_t)J.func, grid.x, grid.y, grid.z, block.x, 1, 1, (unsigned)J.ldsBytes, st, args, nullptr) == hipSuccess;
}

template <int MODE>
static void launch_tile(int G, dim3 grid, dim3 block, size_t ldsBytes, hipStream_t st, const MedProgDev &P, const MedTileArgs &A) {
  switch (G) {
    case 1: hipLaunchKernelGGL((k_medium_tile<MODE, 1>), grid, block, ldsBytes, st, P, A); break;
    case 2: hipLaunchKernelGGL((k_medium_tile<MODE, 2>), grid, block, ldsBytes, st, P, A); break;
    case 4: hipLaunchKernelGGL((k_medium_tile<MODE, 4>), grid, block, ldsBytes, st, P, A); break;
    case 8: hipLaunchKernelGGL((k_medium_tile<MODE, 8>), grid, block, ldsBytes, st, P, A); break;
    case 16: hipLaunchKernelGGL((k_medium_tile<MODE, 16>), grid, block, ldsBytes, st, P, A); break;
    case 32: hipLaunchKernelGGL((k_medium_tile<MODE, 32>), grid, block, ldsBytes, st, P, A); break;
    default: hipLaunchKernelGGL((k_medium_tile<MODE, 64>), grid, block, ldsBytes, st, P, A); break;
  }
}

static bool g_attr_set = false;
static void set_lds_attr() {
  if (g_attr_set) return;
#define SET(M, GG) (void)hipFuncSetAttribute((const void *)k_medium_tile<M, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  SET(0, 1); SET(0, 2); SET(0, 4); SET(0, 8); SET(0, 16); SET(0, 32); SET(0, 64);
  SET(1, 1); SET(1, 2); SET(1, 4); SET(1, 8); SET(1, 16); SET(1, 32); SET(1, 64);
#undef SET
  g_attr_set = true;
}

// Launch the wavefront of parallelogram tiles of a set of pairs whose first launches (PairDesc::launch0) are given:
// tile (pair, strip a, block b) runs in launch launch0 + 2a + b.  Builds the dense per-launch tile lists.
// buffers of a tile sweep that keeps no matrix (MED_MAT_ROLL)
struct MedRoll {
  double *halo = nullptr; const long long *haloBase = nullptr;     // one halo column per strip: [NA][outLen + 1][max(H, 1)]
  double *bound = nullptr; const long long *boundBase = nullptr;   // one boundary record per strip: [NA][NS - 1][C][S]
  unsigned char *tb = nullptr;                                     // MED_MODE_TB: traceback bytes (PairDesc::cellBase = byte offset)
};

static int launch_wavefront(const mb_machine *m, MedProgram &P, const MedProgDev &devIn, const MedGeom &geo, int mode, int TS,
                            const std::vector<PairDesc> &pairs, const PairDesc *d_pairs, const int *d_in, const int *d_out,
                            double *d_pool, double *d_loglike, hipStream_t st, const double *d_poolB = nullptr,
                            double *d_counts = nullptr, const MedEnv &env = MedEnv(), int matKind = MED_MAT_FULL, const MedRoll *roll = nullptr,
                            bool twoStreams = true) {
  const int C = geo.C;
  const long long n = (long long)pairs.size();
  int nLaunch = 0;
  for (const PairDesc &pd : pairs) {
    const int NA = (pd.inLen + C) / C, NB = (pd.outLen + C + TS - 1) / TS;
    nLaunch = std::max(nLaunch, pd.launch0 + 2 * (NA - 1) + NB);
  }
  // With restricted envelopes a tile none of whose cells lies inside its pair's envelope is not launched: the pool was
  // filled with -inf beforehand (fill_chunk), which is what its cells hold and what the neighbouring tiles' prologues read.
  auto liveTiles = [&](const PairDesc &pd, std::vector<char> &live) {
    const int NA = (pd.inLen + C) / C, NB = (pd.outLen + C + TS - 1) / TS;
    const bool clip = geo.env && env.h_start && pd.envBase >= 0;
    live.assign((size_t)NA * NB, clip ? 0 : 1);
    if (!clip) return;
    for (int o = 0; o <= pd.outLen; ++o) {
      const int es = env.h_start[pd.envBase + o], ee = env.h_end[pd.envBase + o];
      if (ee <= es) continue;
      const int fo = P.backward ? pd.outLen - o : o;
      const int g0 = P.backward ? pd.inLen - (ee - 1) : es, g1 = P.backward ? pd.inLen - es : ee - 1;   // sweep-frame columns of the row
      for (int a = g0 / C; a <= g1 / C; ++a) {
        const int c0 = std::max(g0, a * C) - a * C, c1 = std::min(g1, a * C + C - 1) - a * C;
        for (int b = (fo + c0) / TS; b <= (fo + c1) / TS && b < NB; ++b) live[(size_t)a * NB + b] = 1;
      }
    }
  };
  // TWO GROUPS OF PAIRS ON TWO STREAMS.  A launch is one kernel and the next launch waits for it, so a launch whose tiles are
  // not a multiple of the chip's workgroup slots leaves CUs idle in its last round -- config 4's chunk of 21 pairs has 651 tiles
  // on its plateau for 256 one-workgroup-per-CU slots: 3 rounds for the work of 2.5, and less on the ramps (modelled: 74 % of
  // the slots used).  Pairs are independent: dealt in turn to two (MB_MEDIUM_STREAMS) groups whose launches go to streams of
  // their own, the tail of one group's launch overlaps another group's next one.  Same tiles, same order inside a pair, same results.
  constexpr int MAXG = 4;
  hipStream_t sgs[MAXG] = {st, nullptr, nullptr, nullptr};
  int nG = 1;
  {
    static hipStream_t extra[MAXG - 1] = {nullptr, nullptr, nullptr}; static bool tried = false;
    if (!tried) { tried = true; for (int k = 0; k < MAXG - 1; ++k) if (hipStreamCreateWithFlags(&extra[k], hipStreamNonBlocking) != hipSuccess) extra[k] = nullptr; }
    long long peak = 0;
    const int want = std::min(MAXG, env_int_m("MB_MEDIUM_STREAMS", 2));
    if (twoStreams && want > 1) {
      for (const PairDesc &pd : pairs) peak += (pd.inLen + C) / C;                      // tiles of a launch on the plateau: every strip of every pair
      if (peak < 16 * 256)                                                              // (far more tiles than slots: the tail is noise)
        while (nG < want && nG < n && extra[nG - 1]) { sgs[nG] = extra[nG - 1]; ++nG; }
    }
  }
  std::vector<int> cnt((size_t)nG * (nLaunch + 1), 0);        // [group][launch]
  std::vector<char> live;
  for (long long p = 0; p < n; ++p) {
    const PairDesc &pd = pairs[p];
    const int NA = (pd.inLen + C) / C, NB = (pd.outLen + C + TS - 1) / TS;
    liveTiles(pd, live);
    for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) if (live[(size_t)a * NB + b]) cnt[(size_t)(p % nG) * (nLaunch + 1) + pd.launch0 + 2 * a + b] += 1;
  }
  std::vector<long long> off((size_t)nG * (nLaunch + 1) + 1, 0);
  { long long tot = 0; for (size_t k = 0; k < (size_t)nG * (nLaunch + 1); ++k) { off[k] = tot; tot += cnt[k]; } off[(size_t)nG * (nLaunch + 1)] = tot; }
  std::vector<int2> tiles((size_t)off[(size_t)nG * (nLaunch + 1)]);
  {
    std::vector<long long> fill(off.begin(), off.end() - 1);
    for (long long p = 0; p < n; ++p) {
      const PairDesc &pd = pairs[p];
      const int NA = (pd.inLen + C) / C, NB = (pd.outLen + C + TS - 1) / TS;
      liveTiles(pd, live);
      for (int a = 0; a < NA; ++a)
        for (int b = 0; b < NB; ++b)
          if (live[(size_t)a * NB + b])   // (bit 30 of the strip: the block before this one did not run -- a tile without a matrix then starts from -inf instead of its boundary record)
            tiles[(size_t)fill[(size_t)(p % nG) * (nLaunch + 1) + pd.launch0 + 2 * a + b]++] = make_int2((int)p, a | ((matKind == MED_MAT_ROLL && b > 0 && !live[(size_t)a * NB + b - 1]) ? (1 << 30) : 0));
    }
  }
  int2 *d_tiles = nullptr;
  if (!hip_ok(sm_alloc((void **)&d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(int2)), "hipMalloc(tile list)")) return 1;
  if (!tiles.empty() && h2d_large(d_tiles, tiles.data(), tiles.size() * sizeof(int2))) { sm_free(d_tiles); return 1; }   // staged: see h2d_large
  const MedJit *J = medium_jit_get(m, P, geo, mode, matKind) ? &P.jit[medium_jit_slot(mode, matKind, geo.level, geo.env)] : nullptr;
  if ((mode == MED_MODE_COUNT || mode == MED_MODE_TB || matKind == MED_MAT_ROLL || geo.env) && !J) { sm_free(d_tiles); return -1; }   // no ahead-of-time kernel for these
  MedProgDev dev = devIn;
  dev.rec = P.dev.rec; dev.ldsImage = P.dev.ldsImage; dev.ldsImageRecs = (int)P.ldsImageIdx.size();
  if (geo.compact) { dev.rec = P.d_recC; dev.ldsImage = P.d_ldsImageC; }
  MedTileArgs A{};
  A.pairs = d_pairs; A.inTok = d_in; A.outTok = d_out; A.pool = d_pool; A.colHalo = nullptr; A.haloBase = nullptr;
  A.loglike = d_loglike; A.tiles = d_tiles; A.C = C; A.TS = TS; A.rev = P.backward ? 1 : 0; A.materialise = 1;
  A.poolB = d_poolB; A.counts = d_counts; A.envStart = env.d_start; A.envEnd = env.d_end; A.det = g_deterministic ? 1 : 0;
  if (roll) { A.colHalo = roll->halo; A.haloBase = roll->haloBase; A.bound = roll->bound; A.boundBase = roll->boundBase; A.tb = roll->tb; }
  const dim3 block(geo.waves * 64);
  hipEvent_t evStart = nullptr, evDone = nullptr;
  bool streamsOk = true, launchFailed = false;
  if (nG > 1) {      // the other streams start behind what `st` has queued (tile list, buffers) and hand back to it at the end
    streamsOk = hipEventCreateWithFlags(&evStart, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&evDone, hipEventDisableTiming) == hipSuccess &&
                hipEventRecord(evStart, st) == hipSuccess;
    for (int g = 1; g < nG && streamsOk; ++g) streamsOk = hipStreamWaitEvent(sgs[g], evStart, 0) == hipSuccess;
  }
  for (int l = 0; l < nLaunch && streamsOk; ++l)
    for (int g = 0; g < nG; ++g) {
      const size_t k = (size_t)g * (nLaunch + 1) + l;
      if (cnt[k] <= 0) continue;
      ++g_last_launches;
      A.launch = l; A.tileBase = (int)off[k];
      const dim3 grid((unsigned)cnt[k]);
      hipStream_t sg = sgs[g];
      if (J && launch_jit(*J, grid, block, sg, dev, A)) continue;
      // the ahead-of-time interpreter is the twin of the plain matrix kernels only: a failed launch of a specialised kernel without
      // one (count sweep, traceback bytes, tiles without a matrix, envelopes) ends the sweep with an error
      if (mode == MED_MODE_COUNT || mode == MED_MODE_TB || matKind != MED_MAT_FULL || geo.env) { launchFailed = true; streamsOk = false; break; }
      if (mode == MB_VITERBI) launch_tile<MB_VITERBI>(P.G, grid, block, geo.ldsBytes, sg, dev, A);
      else launch_tile<MB_FORWARD>(P.G, grid, block, geo.ldsBytes, sg, dev, A);
    }
  for (int g = 1; g < nG && streamsOk; ++g) streamsOk = hipEventRecord(evDone, sgs[g]) == hipSuccess && hipStreamWaitEvent(st, evDone, 0) == hipSuccess;
  if (!streamsOk) { for (int g = 0; g < nG; ++g) (void)hipStreamSynchronize(sgs[g]); set_error(launchFailed ? "tile sweep: launch of the run-time specialised kernel failed" : "tile sweep: stream synchronisation failed"); }
  if (evStart) (void)hipEventDestroy(evStart);
  if (evDone) (void)hipEventDestroy(evDone);
  const bool ok = streamsOk && hip_ok(hipGetLastError(), "medium tile launch") && hip_ok(hipStreamSynchronize(st), "medium tile kernels");
  sm_free(d_tiles);
  return ok ? 0 : 1;
}

// Steps per parallelogram tile.  Long sweeps take 128 (half the launches, half the tile prologues: +1.5 % on 10 kb
// outputs); short ones keep 64 so that a pair still cuts into enough tiles to fill the wavefront of launches.
// Round 6: a sweep over FEW pairs is bound by its chain of launches, not by the tiles' prologues, and a chain of 64-step tiles is the
// shorter one (482-state E-step: 6 pairs 220 -> 180 ms, 12 pairs 297 -> 280 ms; scripts/c4b_scale_probe.py).
static int tile_steps(int C, size_t nPairs, int maxOut) {
  int TS = std::max(C, (maxOut >= 4096 && nPairs >= (size_t)env_int_m("MB_MEDIUM_TS_LONG_MIN_PAIRS", 32)) ? 128 : 64);
  const char *e = opt_env("MB_MEDIUM_TS");
  if (e && atoi(e) >= C) TS = atoi(e);
  return TS;
}

MedGeom medium_pick_geometry(const MedProgram &P, const MedGeom &geo, const std::vector<PairDesc> &pairs, bool materialise) {
  // Rule fitted to dnapsw (1 kb) and protpsw (400 aa) sweeps: a strip about an eighth of the longest input -- the last
  // strip's padding and the skewed start/end of every sweep then cost ~10 % -- but never fewer than 4 wavefronts per
  // workgroup for the materialised kernels (2 for the rolling one, which has no stores to hide).
  MedGeom best = geo;
  best.level = 0;
  if (pairs.empty() || env_int_m("MB_MEDIUM_ADAPT_WIDTH", 1) == 0) return best;
  int maxIn = 0;
  for (const PairDesc &pd : pairs) maxIn = std::max(maxIn, pd.inLen);
  const int minWaves = std::min(geo.waves, materialise ? 4 : 2);
  int waves = geo.waves;
  for (int h = 1; h < MED_GEOM_LEVELS; ++h) {
    const int w2 = waves / 2;
    if (w2 < minWaves || (long long)w2 * P.G * 8 < maxIn + 1) break;
    waves = w2;
    best.waves = waves; best.C = waves * P.G; best.level = h;
  }
  best.ldsBytes = (size_t)P.NS * (best.C + 1) * P.Spad * sizeof(double);
  return best;
}

// The in-place ring's geometry (MedProgram::inPlaceOk): as many wavefronts as its LDS footprint allows (MB_MEDIUM_COMPACT_MAXWAVES, 12),
// decided ONCE per kernel kind by compiling the kernel: it must fit the registers that wavefront count leaves WITHOUT a re-plan (the
// placement is shared with the program's other kernels; the traceback-byte program, whose only other kernel serves mb_fill, may
// re-plan), else the kind keeps the plain ring.  OFF unless MB_MEDIUM_INPLACE_RING=1: psw2dna's rolling sweeps run 1 128-1 190 G cells/s at 12
// wavefronts against 1 200-1 319 at the plain ring's 8 (9 wavefronts: 1 000) -- a SIMD's time grows with the wavefronts on it.
static long long g_inplace_kernels = 0;
long long medium_inplace_kernels() { return g_inplace_kernels; }
MedGeom medium_roll_geometry(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const std::vector<PairDesc> &pairs, int mode, int matKind, bool env, bool materialiseRule) {
  const int kind = (3 * medium_jit_index(mode) + matKind) * 2 + (env ? 1 : 0);
  auto plain = [&]() { MedGeom g = medium_pick_geometry(P, geoIn, pairs, materialiseRule); g.env = env; g.haloSteps = 0; return g; };
  if (matKind == MED_MAT_FULL || mode != MB_FORWARD || P.counting || P.recC.empty() || !P.d_recC || P.LPG <= 2 || env_int_m("MB_MEDIUM_INPLACE_RING", 0) == 0) return plain();      // (off by default: measured, it does not pay -- DESIGN.md 4.1d)
  if (P.compactState[kind] < 0) return plain();
  if (P.compactState[kind] == 0) {
    P.compactState[kind] = -1;
    const int KC = medium_compact_len(P);
    if ((long long)P.Spad + (long long)P.NS * KC >= (long long)P.NS * P.Spad) return plain();      // nothing to gain (nearly every state is read across columns)
    MedGeom g = geoIn;
    g.compact = true; g.env = env; g.haloSteps = 0; g.level = 0;
    const int maxWaves = std::min(16, env_int_m("MB_MEDIUM_COMPACT_MAXWAVES", 12));
    for (g.waves = maxWaves; g.waves >= geoIn.waves; --g.waves) {      // (the same wavefront count is allowed: the in-place ring also drops an address add per candidate)
      g.C = g.waves * P.G;
      g.ldsBytes = (size_t)(g.C + 1) * (size_t)(P.Spad + P.NS * KC) * sizeof(double);
      if (medium_jit_lds_bytes(P, g, mode) <= 160 * 1024 - 512) break;
    }
    if (g.waves < geoIn.waves) return plain();
    MedJit &J = P.jit[medium_jit_slot(mode, matKind, 0, env)];
    if (J.tried && !J.func) return plain();                                                       // (the kind has no specialised kernel at all)
    if (J.module) (void)hipModuleUnload((hipModule_t)J.module);
    J = MedJit();
    // (9 ... 12 wavefronts leave a wavefront the same 168 registers, 5 ... 8 the same 256: at most two attempts -- the most wavefronts
    //  the LDS allows, then 8 when that is still more than the plain ring's)
    bool ok = medium_jit_get(m, P, g, mode, matKind, /*allowReplan=*/false);
    if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] in-place ring (mode %d, matrix kind %d%s): %zu of %d states per short vector, %d wavefronts -> %s\n", mode, matKind, env ? ", envelopes" : "",
                                                  P.haloStates.size(), m->S, g.waves, ok ? "in use" : "does not fit its registers");
    if (!ok && g.waves > 8 && geoIn.waves < 8) {
      J = MedJit();
      g.waves = 8; g.C = g.waves * P.G;
      g.ldsBytes = (size_t)(g.C + 1) * (size_t)(P.Spad + P.NS * KC) * sizeof(double);
      ok = medium_jit_get(m, P, g, mode, matKind, /*allowReplan=*/false);
      if (opt_env("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] in-place ring: 8 wavefronts -> %s\n", ok ? "in use" : "does not fit its registers: plain ring");
    }
    if (!ok) { J = MedJit(); return plain(); }
    P.compactState[kind] = 1;
    P.compactWaves[kind] = g.waves;
    ++g_inplace_kernels;
  }
  MedGeom base = geoIn;
  base.compact = true; base.waves = P.compactWaves[kind]; base.C = base.waves * P.G;
  MedGeom g = medium_pick_geometry(P, base, pairs, materialiseRule);
  g.compact = true; g.env = env; g.haloSteps = 0;
  g.ldsBytes = (size_t)(g.C + 1) * (size_t)(P.Spad + P.NS * medium_compact_len(P)) * sizeof(double);
  return g;
}

static int max_out_len(const std::vector<PairDesc> &pairs) {
  int mx = 0;
  for (const PairDesc &pd : pairs) mx = std::max(mx, pd.outLen);
  return mx;
}

// Materialised fill of a chunk of pairs whose matrices are all kept (Viterbi, Backward, counts, mb_fill).
int medium_fill_materialised(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, int mode, int startNode,
                             const PairDesc *d_pairs, const std::vector<PairDesc> &pairsIn, const int *d_in, const int *d_out,
                             double *d_pool, hipStream_t st, const MedEnv &env) {
  if (pairsIn.empty()) return 0;
  MedGeom geo = medium_pick_geometry(P, geoIn, pairsIn, true);
  geo.env = env.d_start != nullptr;
  set_lds_attr();
  std::vector<PairDesc> pairs = pairsIn;
  for (PairDesc &pd : pairs) pd.launch0 = 0;
  MedProgDev dev = P.dev;
  if (startNode >= 0 && !P.closure && !P.backward) dev.seedOff = (unsigned)startNode * 8u;   // ForwardMatrix(.., startState)
  return launch_wavefront(m, P, dev, geo, mode, tile_steps(geo.C, pairs.size(), max_out_len(pairs)), pairs, d_pairs, d_in, d_out, d_pool, nullptr, st,
                          nullptr, nullptr, env);
}

int medium_counts_materialised(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const PairDesc *d_pairs,
                               const std::vector<PairDesc> &pairsIn, const int *d_in, const int *d_out, double *d_fwd,
                               const double *d_bwd, double *d_counts, double *d_loglike, hipStream_t st, const MedEnv &env) {
  if (pairsIn.empty()) return 0;
  MedGeom geo = medium_pick_geometry(P, geoIn, pairsIn, true);
  geo.env = env.d_start != nullptr;
  if (!P.counting || !medium_jit_get(m, P, geo, MED_MODE_COUNT, MED_MAT_FULL)) return -1;
  std::vector<PairDesc> pairs = pairsIn;
  for (PairDesc &pd : pairs) pd.launch0 = 0;
  return launch_wavefront(m, P, P.dev, geo, MED_MODE_COUNT, tile_steps(geo.C, pairs.size(), max_out_len(pairs)), pairs, d_pairs, d_in, d_out, d_fwd,
                          d_loglike, st, d_bwd, d_counts, env);
}

// Halo columns and boundary records of a set of pairs for a tile sweep without a matrix, in workspaces 11 / 12.
static int roll_buffers(const MedProgram &P, const MedGeom &geo, const std::vector<PairDesc> &pairs, bool env, hipStream_t st, MedRoll &R,
                        long long **d_bases) {
  const long long n = (long long)pairs.size(), H = std::max<long long>((long long)P.haloStates.size(), 1);
  std::vector<long long> base(2 * n);
  long long haloD = 0, boundD = 0;
  for (long long p = 0; p < n; ++p) {
    const long long NA = (pairs[p].inLen + geo.C) / geo.C;
    base[p] = haloD; base[n + p] = boundD;
    haloD += NA * (pairs[p].outLen + 1) * H;
    boundD += NA * ((P.NS - 1) * geo.C * P.dev.S + (geo.compact ? (long long)(geo.C * (P.dev.S + (P.NS - 1) * medium_compact_len(P))) : 0));      // (in-place ring: the full vectors of the last step + the short ones)
  }
  if (!hip_ok(sm_alloc((void **)d_bases, 2 * n * sizeof(long long)), "hipMalloc(roll bases)")) return 1;
  if (!hip_ok(hipMemcpyAsync(*d_bases, base.data(), 2 * n * sizeof(long long), hipMemcpyHostToDevice, st), "H2D roll bases") ||
      !hip_ok(hipStreamSynchronize(st), "H2D roll bases")) return 1;   // (`base` is a pageable host vector)
  R.halo = (double *)ws_get(11, (size_t)std::max<long long>(haloD, 1) * sizeof(double));
  R.bound = (double *)ws_get(12, (size_t)std::max<long long>(boundD, 1) * sizeof(double));
  if (!R.halo || !R.bound) return 1;
  R.haloBase = *d_bases; R.boundBase = *d_bases + n;
  if (env && launch_fill_neg_inf(R.halo, haloD, st)) return 1;   // halo rows of tiles that do not run
  return 0;
}

// ViterbiMatrix::fill (src/viterbi.cpp:18-43) keeping ONE traceback byte per cell instead of the fp64 cell (1 B per
// cell): tiles without a matrix, scores of the end cells in d_loglike.  pairs[].cellBase = BYTE offset of the pair's bytes in d_tb
// (medium_tb_stride(S) per supercell, reference order).  Returns -1 (nothing launched) when the specialised kernel is unavailable.
int medium_viterbi_tb(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const PairDesc *d_pairs, const std::vector<PairDesc> &pairsIn,
                      const int *d_in, const int *d_out, unsigned char *d_tb, double *d_loglike, hipStream_t st, const MedEnv &env) {
  if (pairsIn.empty()) return 0;
  if (!medium_tb_eligible(m, P)) return -1;
  MedGeom geo = medium_roll_geometry(m, P, geoIn, pairsIn, MED_MODE_TB, MED_MAT_ROLL, env.d_start != nullptr, true);
  // the byte vectors of a step (one per column) need LDS the widest strip does not leave: columns are given up, one wavefront
  // at a time, until they fit (psw2dna: 32 -> 28 columns); the same rule for every batch, so a strip level maps to one kernel
  while (geo.waves > 1 && medium_jit_lds_bytes(P, geo, MED_MODE_TB) > 160 * 1024) { --geo.waves; geo.C = geo.waves * P.G; }
  if (!medium_jit_get(m, P, geo, MED_MODE_TB, MED_MAT_ROLL)) return -1;
  std::vector<PairDesc> pairs = pairsIn;
  for (PairDesc &pd : pairs) pd.launch0 = 0;
  MedRoll R; long long *d_bases = nullptr;
  int rc = roll_buffers(P, geo, pairs, geo.env, st, R, &d_bases);
  R.tb = d_tb;
  if (!rc) rc = launch_wavefront(m, P, P.dev, geo, MED_MODE_TB, tile_steps(geo.C, pairs.size(), max_out_len(pairs)), pairs, d_pairs, d_in, d_out, nullptr,
                                 d_loglike, st, nullptr, nullptr, env, MED_MAT_ROLL, &R);
  sm_free(d_bases);
  return rc;
}

// RollingOutputForwardMatrix semantics (`boss --loglike`: log-likelihoods only) through the TILE pipeline without a matrix: what a
// batch of fewer pairs than CUs wants (one workgroup per pair, the JMAT == 0 sweep, would leave most of the chip idle; the tile
// pipeline with a matrix pays 8 B per cell for nothing), and what an enveloped batch wants (tiles outside the band do not run).
// d_loglike must be pre-filled with -inf when envelopes are present.  Returns -1 when the specialised kernel is unavailable.
int medium_forward_rolltiles(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const PairDesc *d_pairs, const std::vector<PairDesc> &pairsIn,
                             const int *d_in, const int *d_out, double *d_loglike, hipStream_t st, const MedEnv &env) {
  if (pairsIn.empty()) return 0;
  if (P.counting) return -1;
  MedGeom geo = medium_roll_geometry(m, P, geoIn, pairsIn, MB_FORWARD, MED_MAT_ROLL, env.d_start != nullptr, true);
  if (!medium_jit_get(m, P, geo, MB_FORWARD, MED_MAT_ROLL)) {
    if (!geo.compact) return -1;
    geo = medium_pick_geometry(P, geoIn, pairsIn, true); geo.env = env.d_start != nullptr; geo.haloSteps = 0;      // (a narrower strip level whose compact kernel does not build)
    if (!medium_jit_get(m, P, geo, MB_FORWARD, MED_MAT_ROLL)) return -1;
  }
  std::vector<PairDesc> pairs = pairsIn;
  for (PairDesc &pd : pairs) pd.launch0 = 0;
  MedRoll R; long long *d_bases = nullptr;
  int rc = roll_buffers(P, geo, pairs, geo.env, st, R, &d_bases);
  if (!rc) rc = launch_wavefront(m, P, P.dev, geo, MB_FORWARD, tile_steps(geo.C, pairs.size(), max_out_len(pairs)), pairs, d_pairs, d_in, d_out, nullptr,
                                 d_loglike, st, nullptr, nullptr, env, MED_MAT_ROLL, &R);
  sm_free(d_bases);
  return rc;
}

// Forward sweep fused with MachineCounts accumulation that keeps NO Forward matrix (src/backward.cpp:58-87 needs F(i,o,src) only
// while the supercell is in LDS): the Backward matrices are read once, nothing else moves -- 16 B per lattice cell with the
// Backward fill.  Returns -1 when the specialised kernel is unavailable.
int medium_counts_rolling(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const PairDesc *d_pairs, const std::vector<PairDesc> &pairsIn,
                          const int *d_in, const int *d_out, const double *d_bwd, double *d_counts, double *d_loglike, hipStream_t st, const MedEnv &env) {
  if (pairsIn.empty()) return 0;
  MedGeom geo = medium_pick_geometry(P, geoIn, pairsIn, true);
  geo.env = env.d_start != nullptr; geo.haloSteps = 0;
  if (!P.counting || !medium_jit_get(m, P, geo, MED_MODE_COUNT, MED_MAT_ROLL)) return -1;
  std::vector<PairDesc> pairs = pairsIn;
  for (PairDesc &pd : pairs) pd.launch0 = 0;
  MedRoll R; long long *d_bases = nullptr;
  int rc = roll_buffers(P, geo, pairs, geo.env, st, R, &d_bases);
  if (!rc) rc = launch_wavefront(m, P, P.dev, geo, MED_MODE_COUNT, tile_steps(geo.C, pairs.size(), max_out_len(pairs)), pairs, d_pairs, d_in, d_out, nullptr,
                                 d_loglike, st, d_bwd, d_counts, env, MED_MAT_ROLL, &R);
  sm_free(d_bases);
  return rc;
}

// Materialised Forward over a whole batch when only the log-likelihoods are kept (ForwardMatrix(...).logLike()):
// a continuous pipeline.  Pair p starts at launch launch0[p]; a new pair is admitted as soon as the wavefront has
// room for its strips (target: one resident workgroup per CU) and a matrix slot is free, so the chip stays full
// across pair boundaries instead of draining at every sub-batch.  Matrix slots are recycled in stream order.
int medium_forward_pipelined(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const std::vector<PairDesc> &pairsIn,
                             const int *d_in, const int *d_out, double *d_pool, long long poolCells, double *d_loglike,
                             hipStream_t st) {
  const long long n = (long long)pairsIn.size();
  if (n == 0) return 0;
  const MedGeom geo = medium_pick_geometry(P, geoIn, pairsIn, true);
  set_lds_attr();
  const int C = geo.C, S = m->S;
  long long slotCells = 0;
  for (const PairDesc &pd : pairsIn) slotCells = std::max(slotCells, (long long)(pd.inLen + 1) * (pd.outLen + 1) * S);
  const long long nSlots = std::min<long long>(poolCells / std::max<long long>(slotCells, 1), n);
  if (nSlots < 1) { set_error("a single DP matrix exceeds the device memory budget"); return 1; }
  const int TS = tile_steps(C, (size_t)n, max_out_len(pairsIn));
  // resident workgroups the chip can hold: per CU as many as the LDS footprint and the 32-wavefront limit allow
  const size_t ldsPerWg = std::max<size_t>(medium_jit_lds_bytes(P, geo), 1024);
  const int wgPerCU = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((160 * 1024) / ldsPerWg, 32 / std::max(geo.waves, 1)), 8));
  const int target = 256 * wgPerCU;
  std::vector<PairDesc> pairs = pairsIn;
  std::vector<int> life(n), NAp(n), NBp(n);
  for (long long p = 0; p < n; ++p) {
    NAp[p] = (pairs[p].inLen + C) / C;
    NBp[p] = (pairs[p].outLen + C + TS - 1) / TS;
    life[p] = 2 * (NAp[p] - 1) + NBp[p];
  }
  // tiles of a pair that run k launches after its first one (a trapezoid: +1 strip every 2 launches, plateau, drain)
  auto profile = [&](long long p, int k) {
    int cnt = 0;
    for (int a = 0; a < NAp[p]; ++a) { const int b = k - 2 * a; if (b >= 0 && b < NBp[p]) ++cnt; }
    return cnt;
  };
  std::vector<int> load;                           // committed tiles per future launch
  std::vector<long long> slotFreeAt(nSlots, 0);    // launch index from which the slot may be reused
  long long next = 0, firstAlive = 0;
  int launch = 0;
  while (firstAlive < n) {
    // admit the next pair as soon as a matrix slot is free and no launch of its life would exceed the target
    while (next < n) {
      long long slot = -1;
      for (long long k = 0; k < nSlots; ++k) if (slotFreeAt[k] <= launch) { slot = k; break; }
      if (slot < 0) break;
      if ((int)load.size() < launch + life[next]) load.resize(launch + life[next], 0);
      bool fits = true;
      if (next > firstAlive)   // an empty pipeline always admits
        for (int k = 0; k < life[next] && fits; ++k) fits = load[launch + k] + profile(next, k) <= target;
      if (!fits) break;
      for (int k = 0; k < life[next]; ++k) load[launch + k] += profile(next, k);
      pairs[next].launch0 = launch;
      pairs[next].cellBase = slot * slotCells;
      slotFreeAt[slot] = launch + life[next];
      ++next;
    }
    while (firstAlive < next && pairs[firstAlive].launch0 + life[firstAlive] <= launch) ++firstAlive;
    ++launch;
  }
  PairDesc *d_pairs = nullptr;
  if (!hip_ok(sm_alloc((void **)&d_pairs, n * sizeof(PairDesc)), "hipMalloc(pairs)")) return 1;
  if (!hip_ok(hipMemcpyAsync(d_pairs, pairs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, st), "H2D pairs")) { sm_free(d_pairs); return 1; }
  // (one stream: a matrix slot is handed from one pair to the next in stream order)
  const int rc = launch_wavefront(m, P, P.dev, geo, MB_FORWARD, TS, pairs, d_pairs, d_in, d_out, d_pool, d_loglike, st, nullptr, nullptr, MedEnv(), MED_MAT_FULL, nullptr, false);
  sm_free(d_pairs);
  return rc;
}

// Rolling (log-likelihood only) Forward: one workgroup per pair per launch, strips in sequence.
int medium_forward_rolling(const mb_machine *m, MedProgram &P, const MedGeom &geoIn, const PairDesc *d_pairs,
                           const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_colHalo,
                           const long long *d_haloBase, double *d_loglike, hipStream_t st) {
  if (pairs.empty()) return 0;
  MedGeom geo = medium_roll_geometry(m, P, geoIn, pairs, MB_FORWARD, MED_MAT_NONE, false, false);
  if (geo.compact && !medium_jit_get(m, P, geo, MB_FORWARD, MED_MAT_NONE)) { geo = medium_pick_geometry(P, geoIn, pairs, false); }      // (the ahead-of-time twin knows the plain ring only)
  set_lds_attr();
  int maxIn = 0, maxOut = 0;
  for (const PairDesc &pd : pairs) { maxIn = std::max(maxIn, pd.inLen); maxOut = std::max(maxOut, pd.outLen); }
  const int C = geo.C, NA = (maxIn + C) / C;
  MedTileArgs A{};
  A.pairs = d_pairs; A.inTok = d_in; A.outTok = d_out; A.pool = nullptr; A.colHalo = d_colHalo; A.haloBase = d_haloBase;
  A.loglike = d_loglike; A.C = C; A.TS = maxOut + C + 1; A.rev = 0; A.materialise = 0; A.tiles = nullptr; A.tileBase = 0;
  const dim3 grid((unsigned)pairs.size()), block(geo.waves * 64);
  const MedJit *J = medium_jit_get(m, P, geo, MB_FORWARD, MED_MAT_NONE) ? &P.jit[medium_jit_slot(MB_FORWARD, MED_MAT_NONE, geo.level)] : nullptr;
  MedProgDev dev = P.dev;
  dev.ldsImageRecs = (int)P.ldsImageIdx.size();
  if (geo.compact) { dev.rec = P.d_recC; dev.ldsImage = P.d_ldsImageC; }
  for (int a = 0; a < NA; ++a) {
    A.launch = a;
    ++g_last_launches;
    if (J && launch_jit(*J, grid, block, st, dev, A)) continue;
    launch_tile<MB_FORWARD>(P.G, grid, block, geo.ldsBytes, st, dev, A);
  }
  return hip_ok(hipGetLastError(), "medium rolling launch") ? 0 : 1;
}

}  // namespace mb

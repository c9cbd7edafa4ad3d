// mb_medium_jit_src.h -- source skeleton of the run-time specialised "lanes = states" tile kernel.
//
// The ahead-of-time kernel in mb_medium.hip interprets a compiled program (descriptors + switch on slot counts); its
// instruction stream is dominated by that interpretation.  Here the same program is unrolled at run time into
// straight-line HIP for ONE machine and compiled with hiprtc for gfx950: slot counts, record offsets, which LDS vector a
// candidate reads, the state count and the strip geometry are all literals; candidate records that do not change along a
// column's sweep sit in VGPRs, output-token dependent ones and the sweep's output tokens in LDS (placement: see
// mb_medium_jit.cpp), so a step issues no global load besides the halo supercell.  Semantics are identical to k_medium_tile
// (tests run both and compare); if hiprtc is unavailable the engine silently keeps the ahead-of-time kernel.
//
// Markers replaced by the generator (mb_medium_jit.cpp):  /*@DEFS@*/  /*@PRE@*/  /*@BODY@*/  /*@POST@*/
#pragma once

namespace mb {
static const char *kMedJitSkeleton = R"MBJIT(
/*@DEFS@*/
#define NEG_INF (-__builtin_inf())
#define MED_L2E 1.44269504088896f
#define MED_LN2 0.693147180559945f
struct PairDesc { long long inBase, outBase; int inLen, outLen; long long cellBase; int launch0; int pad; long long envBase; };
struct MedRec { double w; unsigned srcOff; unsigned dstOff; };
struct MedProgDev {
  int S, Spad, LPG, G, NS, nChunks;
  int nIn, nOut, startNode, endNode;
  unsigned seedOff;
  int ldsImageRecs;
  const int *desc;
  const MedRec *rec;
  const MedRec *ldsImage;
  const int *accMap;
};
struct MedTileArgs {
  const PairDesc *pairs;
  const int *inTok, *outTok;
  double *pool;
  double *colHalo;
  const long long *haloBase;
  double *loglike;
  const int2 *tiles;
  int C, TS, launch, rev, materialise, tileBase, det;      // det: counts in 64-bit fixed point (MB_DETERMINISTIC, mb_internal.h)
  const double *poolB;
  double *counts;
  const int *envStart, *envEnd;
  double *bound;                // JMAT == 2: tile-boundary records (ring state a block hands to the next block of its strip)
  const long long *boundBase;   //            per pair offset (doubles); strip a's record follows at a * (JNS - 1) * JC * JS
  unsigned char *tb;            // JTB: Viterbi traceback bytes, JSB per supercell, reference order; PairDesc::cellBase = byte offset
};
// JMAT: 0 = one workgroup sweeps a whole strip, no matrix (halo columns in colHalo); 1 = tiles, matrix in `pool` (the ring
// state of a tile's first steps and the halo supercells are read back from it); 2 = tiles WITHOUT a matrix: halo columns
// (JNH states per row: the sources of input-consuming transitions) in colHalo, one column per strip, and the ring state at
// a block boundary in `bound` -- what Viterbi with traceback bytes and the count sweep use (nothing but the bytes / the
// Backward matrix moves through HBM).
#define JTILES (JMAT != 0)
#if JENV
#define JCLIP(x) (inside ? (x) : NEG_INF)      // cells outside the pair's envelope stay -inf (src/dpmatrix.defs.h:36, dpmatrix.h:142-144)
#define JINSIDE inside
#else
#define JCLIP(x) (x)
#define JINSIDE active
#endif
#if JMODE == 2
#define SRCOFF(x) ((int)((x) & 0xFFFFu))
#define DSTOFF(x) ((x) & 0xFFFFu)      // (upper half of a fused emit round's slot-0 record: the real state's place in the Backward supercell)
#elif JCR
#define SRCOFF(x) ((int)(x))
#define DSTOFF(x) ((x) & 0xFFFFu)      // (upper half: 8 x (the state's place in the column's short vector + 1), 0 = not one of its states)
#else
#define SRCOFF(x) ((int)(x))
#define DSTOFF(x) (x)
#endif
// IN-PLACE RING (JCR, mb_medium.h MedProgram::inPlaceOk): per column ONE full vector -- the cells of the step before when the emit rounds
// of stage 0 read it (aDown = aCur: all their loads are issued before the stage's first store), this step's cells afterwards -- and JNS
// short vectors of the halo states (JNH: the sources of input-consuming candidates, all a neighbouring column ever reads):
// [cells: JSPAD][short 0: JKC] ... [short JNS-1: JKC]; step t writes short vector t % JNS beside the full one, the column to the right
// reads (t - 1) % JNS (aLeft) and (t - 2) % JNS (aDiag)
#if JCR
#define JCSTORE(d, v) do { if ((d) >> 16) *(double *)(ldsb + (aCompW + (int)((d) >> 16) - 8)) = (v); } while (0)
#else
#define JCSTORE(d, v) do { } while (0)
#endif
typedef const __attribute__((address_space(4))) int *cdesc_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef const __attribute__((address_space(1))) u32x4 *grec_t;
typedef const __attribute__((address_space(1))) char *gbytes_t;
struct Rec { double w; unsigned srcOff, dstOff; };

__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }   // v_max_f64; operands are never NaN
__device__ __forceinline__ Rec mk_rec(u32x4 r) {
  Rec o; o.w = __hiloint2double((int)r.y, (int)r.x); o.srcOff = r.z; o.dstOff = r.w; return o;
}
__device__ __forceinline__ Rec ld_g(gbytes_t base, unsigned off) { return mk_rec(*(grec_t)(base + off)); }
__device__ __forceinline__ Rec ld_b(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  return mk_rec(__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, 0));
}
__device__ __forceinline__ Rec ld_l(const char *base, unsigned off) { return mk_rec(*(const u32x4 *)(base + off)); }
__device__ __forceinline__ double med_lds(const char *ldsb, int off) { return *(const double *)(ldsb + off); }
__device__ __forceinline__ float ex2(double d) { return __builtin_amdgcn_exp2f((float)d * MED_L2E); }
// count mode: add exp(x) to the LDS accumulator at byte offset off (ds_add_f64; lanes of one column never collide)
// deterministic mode (jdet = MedTileArgs::det in scope): the LDS accumulators hold 64-bit fixed point at 2^-44 -- integer adds commute
#define cnt_flush(l, o, a) cnt_flush_(l, o, a, jdet)
__device__ __forceinline__ void cnt_flush_(const char *ldsb, unsigned off, float acc, int jdet) {
  // (clamped to [0, 2^62] before the cast: a NaN, a negative or a huge term saturates instead of being undefined behaviour, and the host
  //  reads an accumulator >= 2^62 as "overflowed" -- ADVICE r4)
  if (jdet) (void)__hip_atomic_fetch_add((unsigned long long *)(ldsb + off), (unsigned long long)__builtin_fmin(__builtin_fmax((double)acc * 17592186044416.0, 0.0), 4611686018427387904.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else (void)__hip_atomic_fetch_add((double *)(ldsb + off), (double)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define cnt_add(l, o, x) cnt_flush_(l, o, ex2(x), jdet)
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
__device__ __forceinline__ int med_vofs(int vsel, int sCur, int sPrev, int sPrev2, int colStride) {
  const int slotOff = vsel == 3 ? sCur : (vsel == 0 ? sPrev2 : sPrev);
  return slotOff - (vsel < 2 ? colStride : 0);
}

// finished supercell: LDS -> global, the LPG lanes of the column side by side.  JSTORE2: 16 bytes per lane and store
// (global_store_dwordx4; the destination is only 8-byte aligned when the state count is odd, which global memory allows)
typedef double d2a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void med_copy_out(double *dstp, const double *cur, int q) {
  constexpr int LPG = 64 / JG, S = JS;
#if JSTORE2
#pragma unroll
  for (int j0 = 0; 2 * j0 < S; j0 += LPG) {
    const int j = 2 * (j0 + q);
    if (j + 1 < S) { d2a8 v; v.x = cur[j]; v.y = cur[j + 1]; *(d2a8 *)(dstp + j) = v; }
    else if (j < S) dstp[j] = cur[j];
  }
#elif JSTORENT
#pragma unroll
  for (int j0 = 0; j0 < S; j0 += LPG) { const int j = j0 + q; if (j < S) __builtin_nontemporal_store(cur[j], dstp + j); }
#else
#pragma unroll
  for (int j0 = 0; j0 < S; j0 += LPG) { const int j = j0 + q; if (j < S) dstp[j] = cur[j]; }
#endif
}

// generic evaluation of one supercell from the descriptors (origin supercell only; same as the AOT slow path)
__device__ __noinline__ void med_slow_supercell(cdesc_t desc, grec_t grec, int nChunks, const char *ldsb, int aDiag, int aLeft, int aDown, int aCur, int aCompW,
                                                int it, int ot, int q,
                                                unsigned seedOff, bool origin, bool lanesOn, int aB, unsigned accBase, double negLL, int tbCol, int jdet) {
  (void)aCompW;
  double accM = NEG_INF; float accS = 0.0f; (void)jdet;
  unsigned code = 0u, prevT = 99u, jT = 0u;   // JTB: (table << 6 | index in the table's list) of the first maximal candidate
  (void)code; (void)prevT; (void)jT; (void)tbCol;
  for (int ch = 0; ch < nChunks; ++ch) {
    cdesc_t dp = desc + ch * 8;
    const int hdr = dp[0];
    const int ns = hdr & 15;
    const bool first = (hdr >> 4) & 1, last = (hdr >> 5) & 1, sync = (hdr >> 6) & 1;
    unsigned dstOff = 0xFFFFFFFFu, dstW = 0u;
    const unsigned vsel = (unsigned)dp[3] >> 24;
    const int vecBase = vsel == 3 ? aCur : (vsel == 0 ? aDiag : (vsel == 1 ? aLeft : aDown));
    const int idx0 = (int)__umul24(it, dp[2]) + (int)__umul24(ot, dp[3] & 0xFFFFFF) + dp[1] + q;
    const unsigned Tsel = (unsigned)dp[3] >> 24;
    if (first || Tsel != prevT) jT = 0u;
    prevT = Tsel;
    for (int k = 0; k < ns; ++k) {
      const Rec r = mk_rec(grec[idx0 + k * dp[4]]);
      bool firstCand = false;
      if (k == 0) {
        dstOff = r.dstOff == 0xFFFFFFFFu ? r.dstOff : DSTOFF(r.dstOff); dstW = r.dstOff;
        if (first) {
          const bool seed = origin && dstOff == seedOff;
          accM = seed ? 0.0 : NEG_INF; accS = seed ? 1.0f : 0.0f;
          firstCand = true;
        }
      }
      const double v = med_lds(ldsb, vecBase + SRCOFF(r.srcOff)) + r.w;
#if JTB
      { const bool take = firstCand || v > accM; code = take ? ((Tsel << 6) | jT) : code; }   // strict >: the FIRST maximum, as std::max_element (src/dpmatrix.defs.h:171-174)
      ++jT;
#endif
      (void)firstCand;
#if JMODE == 2 && !JFLAT
      {  // every chunk's slot-0 record names the lane's destination state
        const double bl = (lanesOn && (int)dstOff >= 0) ? (med_lds(ldsb, aB + (int)dstOff) + negLL) : NEG_INF;
        cnt_add(ldsb, accBase + (r.srcOff >> 16), v + bl);
      }
#endif
      if (JMODE == 1) accM = dmax(accM, v);
      else {
        const double nm = dmax(accM, v), gM = (nm == NEG_INF) ? 0.0 : nm;
        accS = accS * ex2(accM - gM) + ex2(v - gM);
        accM = nm;
      }
    }
    if (last) {
      const double res = (JMODE == 1) ? accM : ((accM == NEG_INF) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);
      if (lanesOn && (int)dstOff >= 0) { *(double *)(ldsb + (aCur + (int)dstOff)) = res; JCSTORE(dstW, res); }
#if JTB
      if (lanesOn && (int)dstOff >= 0) *(unsigned char *)(ldsb + (tbCol + (int)(dstOff >> 3))) = (unsigned char)code;
#endif
    }
    if (sync) med_wave_sync();
  }
}

extern "C" __global__ __launch_bounds__(JWAVES * 64) void k_medium_jit(MedProgDev P, MedTileArgs A) {
  extern __shared__ double lds[];
  constexpr int LPG = 64 / JG, S = JS, Spad = JSPAD, NS = JNS, C = JC, W = JTOKW, NT = JWAVES * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane / LPG, q = lane - g * LPG;
  const int jdet = A.det; (void)jdet;
  int pairIdx, a;
  bool prevDead = false;      // JMAT == 2 with envelopes: the block before this one holds no cell of the envelope and did not run
  if (JTILES) { const int2 tl = A.tiles[A.tileBase + blockIdx.x]; pairIdx = tl.x; a = tl.y & 0x3fffffff; prevDead = (tl.y >> 30) & 1; }
  else { pairIdx = blockIdx.x; a = A.launch; }
  (void)prevDead;
  const PairDesc pd = A.pairs[pairIdx];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const long long I = inLen + 1;
  const int b = JTILES ? A.launch - pd.launch0 - 2 * a : 0;
  const int NA = (inLen + C) / C;
  const int T = outLen + C;
  if (a >= NA || b < 0 || (long long)b * A.TS >= T) return;
  const int t0 = b * A.TS, t1 = min(t0 + A.TS, T);
  const int i0 = a * C;
  const int c = wv * JG + g;
  const int i = i0 + c;
  const bool colValid = (c < C) && (i <= inLen);
  const int *in = A.inTok + pd.inBase, *out = A.outTok + pd.outBase;
  const int rev = A.rev;
  const int it = (colValid && i > 0) ? (rev ? in[inLen - i] : in[i - 1]) : 0;
  double *cells = JMAT == 1 ? A.pool + pd.cellBase : nullptr;
  double *haloIn = nullptr, *haloOut = nullptr;
  if (JMAT == 0) {
    double *hb = A.colHalo + A.haloBase[pairIdx];
    const long long hsz = (long long)(outLen + 1) * S;
    haloIn = hb + ((a + 1) & 1) * hsz;
    haloOut = hb + (a & 1) * hsz;
  }
#if JMAT == 2
  {   // one halo column per strip: row o holds the JNH states other strips read, written by the strip's last column
    double *hb = A.colHalo + A.haloBase[pairIdx];
    const long long hsz = (long long)(outLen + 1) * JNHP;
    haloIn = hb + (long long)max(a - 1, 0) * hsz;
    haloOut = hb + (long long)a * hsz;
  }
#if JCR
  double *bnd = A.bound + A.boundBase[pairIdx] + (long long)a * ((NS - 1) * C * S + C * (S + (NS - 1) * JKC));      // (in-place ring: full vectors of the last step + the short ones; sized by roll_buffers)
#else
  double *bnd = A.bound + A.boundBase[pairIdx] + (long long)a * ((NS - 1) * C * S);
#endif
  // the states this thread moves between a halo row and column 0 of the ring: entries tid, tid + NT, ... of the row
  int myHS[JNHR];
#pragma unroll
  for (int k = 0; k < JNHR; ++k) myHS[k] = JHSTATE(min(tid + k * NT, JNHP - 1));
  (void)myHS;
#endif
#if JTB
  unsigned char *tbPair = A.tb + pd.cellBase;
#endif
  auto cellPtr = [&](int ci, int co) -> double * {
    const long long ri = rev ? inLen - ci : ci, ro = rev ? outLen - co : co;
    return cells + (ro * I + ri) * S;
  };
#if JCR
  constexpr int KCV = JKC, COLD = Spad + NS * JKC;      // doubles per short vector / per column
  auto curOf = [&](int col) -> double * { return lds + (long long)col * COLD; };                              // ring column `col` (0 = the strip's left neighbour): this step's cells
  auto shortOf = [&](int slot, int col) -> double * { return lds + (long long)col * COLD + Spad + slot * KCV; };   // ... its short vector of step parity `slot`
#else
  auto ring = [&](int slot, int col) -> double * { return lds + ((long long)slot * (C + 1) + col) * Spad; };
#endif
  char *ldsRec = (char *)(lds + JRINGD);   // 16-byte aligned
  // output tokens of the sweep, kept in LDS one window of W steps at a time (double buffered): window k holds the
  // tokens of o in [k*W - C + 1, k*W + W - 1]; the token of column c at step t sits at index (t % W) + (C - 1 - c).
  int *tokWin = (int *)(ldsRec + (long long)JLDSRECS * 16);
  int *envSW = tokWin + 2 * (W + C), *envEW = envSW + 2 * (W + C);   // envelope rows (inStart, inEnd) of the same windows (JENV)
  (void)envSW; (void)envEW;
#if JTB
  unsigned char *tbL = (unsigned char *)((((unsigned long long)(tokWin + 6 * (W + C))) + 15ull) & ~15ull);   // traceback bytes of this step's supercells: [C][JTBS], 16-byte aligned
  for (int j = tid; j < C * JTBS / 4; j += NT) ((unsigned int *)tbL)[j] = 0u;
#endif
#if JMODE == 2
  // count mode: the Backward supercell of every column (this step's) and the count accumulators of the workgroup
  double *bvec = (double *)(tokWin + 6 * (W + C));
  double *accL = (double *)((char *)lds + JACCOFF);      // the loop-time accumulators: behind the Backward supercells, and behind what the after-the-loop table will cover
  const double *cellsB = A.poolB + pd.cellBase;
  auto cellPtrB = [&](int ci, int co) -> const double * { return cellsB + ((long long)co * I + ci) * S; };
  const double bLL = cellsB[0];                                   // BackwardMatrix::logLike() = cell(0,0,start), src/backward.cpp:48-50,66
  const double negLL = (bLL > NEG_INF) ? -bLL : NEG_INF;          // -inf likelihood: every term becomes exp(-inf) = 0
  for (int j = tid; j < JNACC; j += NT) accL[j] = 0.0;
#endif
  auto tokAt = [&](int o) -> int { return (o >= 1 && o <= outLen) ? (rev ? out[outLen - o] : out[o - 1]) : 0; };
#if JENV
  // rows of the pair's envelope by (sweep-frame) output position; a pair of the batch without one has [0, inLen + 1)
  const int *envS = pd.envBase >= 0 ? A.envStart + pd.envBase : nullptr, *envE = pd.envBase >= 0 ? A.envEnd + pd.envBase : nullptr;
  auto envSAt = [&](int o) -> int { return (envS && o >= 0 && o <= outLen) ? envS[rev ? outLen - o : o] : 0; };
  auto envEAt = [&](int o) -> int { return (envE && o >= 0 && o <= outLen) ? envE[rev ? outLen - o : o] : inLen + 1; };
  const int iOrig = rev ? inLen - i : i;
#endif
#if JHALOT > 0
  // halo supercells (i0-1, t+1) of ALL the tile's steps, fetched here: the step loop then issues no vector-memory load
#if JMODE == 2
  double *haloBuf = accL + JNACC;
#else
  double *haloBuf = (double *)(tokWin + 6 * (W + C));
#endif
  if (i0 > 0)
    for (int idx = tid; idx < (t1 - t0) * S; idx += NT) {
      const int k = idx / S, j = idx - k * S, ho = t0 + k + 1;
      if (ho <= outLen) haloBuf[idx] = cellPtr(i0 - 1, ho)[j];
    }
#endif

  for (int j = tid; j < JRINGD; j += NT) lds[j] = NEG_INF;
#if JNBSYNC
  // NEIGHBOUR SYNCHRONISATION.  A step of column c reads what column c - 1 held one step earlier, and overwrites the ring slot
  // column c + 1 read one step earlier: a wavefront depends on its two NEIGHBOUR wavefronts only (and the halo rows are moved by
  // wavefront 0 alone), so the workgroup barrier that ended every step -- all wavefronts of a SIMD then stall and issue in
  // lock-step: the vector ALUs ran at a quarter of their issue rate (profiles/r04_valu_model.json, r04_counts4_pmc_sq.txt) -- is
  // replaced by one step counter per wavefront in LDS: wavefront w starts step t once w - 1 and w + 1 have finished step t - 1.
  // All wavefronts of a workgroup are resident, the dependencies run along a chain: the slowest wavefront can always proceed.
  // A real barrier remains at the end of every token window (its double buffer is refilled co-operatively) and of the tile.
  volatile int *stepDone = (volatile int *)((char *)lds + JFLAGOFF);
  if (tid < JWAVES) stepDone[tid] = 0;
#endif
  {  // candidate records placed in LDS (16 B each)
    const u32x4 *img = (const u32x4 *)P.ldsImage;
    u32x4 *dst = (u32x4 *)ldsRec;
    for (int j = tid; j < JLDSRECS; j += NT) dst[j] = img[j];
  }
  {
    const int k0 = t0 / W;
    for (int j = tid; j < W + C - 1; j += NT) tokWin[(k0 & 1) * (W + C) + j] = tokAt(k0 * W - C + 1 + j);
#if JENV
    for (int j = tid; j < W + C - 1; j += NT) { envSW[(k0 & 1) * (W + C) + j] = envSAt(k0 * W - C + 1 + j); envEW[(k0 & 1) * (W + C) + j] = envEAt(k0 * W - C + 1 + j); }
#endif
  }
  __syncthreads();
  // ring state of steps t0-1 (and t0-2 when match edges exist); flat index over (column, state) so that machines with
  // few states still use every thread
#if JCR
  // in-place ring: what earlier steps hand over is, per column of the strip, the full vector of step t0 - 1 and the short vectors of
  // steps t0 - 1 (and t0 - 2) -- JMAT 2: from the boundary record of block b - 1 --, and for column 0 (the strip to the left) the short
  // vectors from its halo column
#if JMAT == 2
  if (b > 0 && !prevDead) {
    for (int idx = tid; idx < C * S; idx += NT) { const int cc = idx / S, j = idx - cc * S; curOf(cc + 1)[j] = bnd[idx]; }
    for (int dt = 1; dt < NS; ++dt) {
      const int slot = (((t0 - dt) % NS) + NS) % NS;
      for (int idx = tid; idx < C * JNHP; idx += NT) { const int cc = idx / JNHP, k = idx - cc * JNHP; shortOf(slot, cc + 1)[k] = bnd[(long long)C * S + (long long)(dt - 1) * C * JNHP + idx]; }
    }
  }
#endif
  for (int dt = 1; dt < NS; ++dt) {
    const int tp = t0 - dt;
    const int slot = ((tp % NS) + NS) % NS;
    const int co = tp + 1;      // ring column 0 is one step ahead: at step tp it holds output position tp + 1
    if (i0 > 0 && co >= 0 && co <= outLen) {
#if JMAT == 0
      for (int k = tid; k < JNH; k += NT) shortOf(slot, 0)[k] = haloIn[(long long)co * S + JHSTATE(k)];
#else
      for (int k = tid; k < JNH; k += NT) shortOf(slot, 0)[k] = haloIn[(long long)co * JNHP + k];
#endif
    }
  }
#else
  for (int dt = 1; dt < NS; ++dt) {
    const int tp = t0 - dt;
    const int slot = ((tp % NS) + NS) % NS;
    for (int idx = tid; idx < (C + 1) * S; idx += NT) {
      const int col = idx / S, j = idx - col * S;
      const int cc = col - 1, ci = i0 + cc, co = tp - cc;
      if (ci < 0 || ci > inLen || co < 0 || co > outLen) continue;
      const double *src = nullptr;
      if (JMAT == 1) src = cellPtr(ci, co);
      else if (JMAT == 0 && cc == -1) src = haloIn + (long long)co * S;
      if (src) ring(slot, col)[j] = src[j];
    }
#if JMAT == 2
    // columns of the strip: the boundary record block b - 1 left behind (cells outside the lattice hold what the ring held:
    // never read by a cell inside it); column 0: the halo rows of steps t0 - 1 (and t0 - 2)
    if (b > 0 && !prevDead)
      for (int idx = tid; idx < C * S; idx += NT) {
        const int cc = idx / S, j = idx - cc * S;
        ring(slot, cc + 1)[j] = bnd[(long long)(dt - 1) * C * S + idx];
      }
    {
      const int co = tp + 1;
      if (i0 > 0 && co >= 0 && co <= outLen) {
#pragma unroll
        for (int k = 0; k < JNHR; ++k) if (tid + k * NT < JNH) ring(slot, 0)[myHS[k]] = haloIn[(long long)co * JNHP + tid + k * NT];
      }
    }
#endif
  }
#endif
  __syncthreads();

  cdesc_t desc = (cdesc_t)P.desc;
  grec_t grec = (grec_t)P.rec;
  gbytes_t grb = (gbytes_t)P.rec;
  const char *ldsb = (const char *)lds;
#if JCR
  constexpr int colStride = COLD * 8, slotStride = 0;      // compact ring: column-major, a column's vectors side by side
#else
  constexpr int colStride = Spad * 8, slotStride = (C + 1) * colStride;
#endif
  (void)slotStride;
  const int myColBase = (c + 1) * colStride;
  const unsigned q16 = (unsigned)q * 16u;
  const unsigned itOff16 = (unsigned)(it * LPG + q) * 16u;
  const __amdgpu_buffer_rsrc_t recRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.rec, 0, 0x7fffffff, 0x00020000);
  (void)q16; (void)itOff16; (void)grb; (void)recRsrc;
#if JMODE == 2
  constexpr int JBV = (S + LPG - 1) / LPG;
  const int aB = (int)((const char *)bvec - ldsb) + c * colStride;
  const unsigned accBase = (unsigned)((const char *)accL - ldsb);
  if (q == 0) bvec[c * Spad + JDUMMYOFF / 8] = NEG_INF;   // read by inactive columns: their terms become exp(-inf) = 0
  {  // Backward supercell of the first step
    const int o0 = min(max(t0 - c, 0), outLen);
    const double *bs = cellPtrB(min(i, inLen), o0);
#pragma unroll
    for (int k = 0; k < JBV; ++k) { const int j = k * LPG + q; if (j < S) bvec[c * Spad + j] = JFLAT ? bs[j] + negLL : bs[j]; }   // JFLAT: bvec holds B - logLike
  }
#else
  constexpr int aB = 0; constexpr unsigned accBase = 0; constexpr double negLL = 0.0;
  (void)aB; (void)accBase; (void)negLL;
#endif
#if JMODE == 2 && JBDIST == 2
  // MB_JIT_B_DISTANCE=2 (experiment, round 4): the Backward supercells fetched TWO steps ahead -- bnext holds B(o + 1) across the
  // step, the loads of B(o + 2) are issued at its top, and the step ends with bvec <- bnext, bnext <- what arrived.  No gain: what
  // the loads cost the count sweep (a fifth of it, MB_JIT_DEBUG experiment in DESIGN.md 4.1c) is their issue, not their latency.
  double bnext[JBV];
  {
    const double *bs = cellPtrB(min(i, inLen), min(max(t0 - c + 1, 0), outLen));
#pragma unroll
    for (int k = 0; k < JBV; ++k) bnext[k] = bs[min(k * LPG + q, S - 1)];
  }
#endif
#if JTB
  const int tbColOff = (int)((const char *)tbL - ldsb) + c * JTBS;
#else
  constexpr int tbColOff = 0;
#endif
  // ---- loop-invariant candidate records (placement REG): one load per sweep, kept in VGPRs --------------------------
/*@PRE@*/
  int slotCur = t0 % NS;
  for (int t = t0; t < t1; ++t) {
    const int o = t - c;
    const bool active = colValid && o >= 0 && o <= outLen;
    const int kw = t / W, tw = t - kw * W;
    const int ot = tokWin[(kw & 1) * (W + C) + tw + (C - 1 - c)];
#if JENV
    const bool inside = active && iOrig >= envSW[(kw & 1) * (W + C) + tw + (C - 1 - c)] && iOrig < envEW[(kw & 1) * (W + C) + tw + (C - 1 - c)];
#endif
    // prefetch the next token window (registers now, LDS at the end of the step)
    int tokPre[JTOKN];
#if JENV
    int envSPre[JTOKN], envEPre[JTOKN];
#endif
    const bool wantTok = (tw == 0 || t == t0) && (kw + 1) * W < t1;
    if (wantTok) {
#pragma unroll
      for (int k = 0; k < JTOKN; ++k) {
        const int j = tid + k * NT;
        tokPre[k] = (j < W + C - 1) ? tokAt((kw + 1) * W - C + 1 + j) : 0;
#if JENV
        envSPre[k] = (j < W + C - 1) ? envSAt((kw + 1) * W - C + 1 + j) : 0;
        envEPre[k] = (j < W + C - 1) ? envEAt((kw + 1) * W - C + 1 + j) : 0;
#endif
      }
    }
    const int slotPrev = (slotCur + NS - 1) % NS, slotPrev2 = (slotCur + NS - 2) % NS;
#if !JCR
    const int sCur = slotCur * slotStride, sPrev = slotPrev * slotStride, sPrev2 = slotPrev2 * slotStride;
#endif
#if JNBSYNC
    {  // both neighbours have finished the step before this one (counters hold the number of steps finished since t0)
      const int need = t - t0;
      if (wv > 0) while (__builtin_amdgcn_readfirstlane(stepDone[wv - 1]) < need) __builtin_amdgcn_s_sleep(1);
      if (wv + 1 < JWAVES) while (__builtin_amdgcn_readfirstlane(stepDone[wv + 1]) < need) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
#endif
    // halo supercell (i0-1, t+1) for the next step.  The load is UNCONDITIONAL (clamped to a valid address when there
    // is no halo): a conditionally initialised register would make the compiler wait for every outstanding memory
    // operation -- the previous step's stores included -- before overwriting it.
    const bool wantHalo = (i0 > 0) && (t + 1 <= outLen);
#if JMAT == 2
    double hvr[JNHR];
#pragma unroll
    for (int k = 0; k < JNHR; ++k) hvr[k] = (JDBG & 2) ? -1.0 : haloIn[(long long)min(t + 1, outLen) * JNHP + min(tid + k * NT, JNHP - 1)];
#elif JCR
    double hvc[JNHR];      // (JMAT 0, in-place ring: only the halo states of the halo supercell)
    {
      const double *hs = haloIn + (long long)min(t + 1, outLen) * S;
#pragma unroll
      for (int k = 0; k < JNHR; ++k) hvc[k] = hs[JHSTATE(min(tid + k * NT, JNHP - 1))];
    }
#elif JHALOT == 0
    double hv[JHALO];
    {
      const int hi = i0 > 0 ? i0 - 1 : 0, ho = min(t + 1, outLen);
      const double *hs = JMAT ? cellPtr(hi, ho) : haloIn + (long long)ho * S;
#pragma unroll
      for (int k = 0; k < JHALO; ++k) hv[k] = hs[min(tid + k * NT, S - 1)];
    }
#endif
#if JMODE == 2
    // Backward supercell (i, o+1) of the next step: registers now, LDS after this step's counting (clamped, unconditional)
    double bpre[JBV];
    {
      const double *bs = cellPtrB(min(i, inLen), min(max(o + JBDIST, 0), outLen));
#pragma unroll
      for (int k = 0; k < JBV; ++k) bpre[k] = (JDBG & 1) ? -1.0 : bs[min(k * LPG + q, S - 1)];
    }
#endif
    // LDS byte addresses of the four vectors this lane's column reads / writes
#if JCR
    const int aCur = myColBase, aCompW = myColBase + (Spad + slotCur * KCV) * 8, aDown = aCur;      // (in place: the emit rounds read the cells of the step before out of the vector they overwrite)
    const int aLeft = myColBase - colStride + (Spad + slotPrev * KCV) * 8, aDiag = myColBase - colStride + (Spad + slotPrev2 * KCV) * 8;
    (void)aCompW;
#else
    const int aCur = myColBase + sCur, aDown = myColBase + sPrev, aLeft = aDown - colStride, aDiag = myColBase + sPrev2 - colStride;
    constexpr int aCompW = 0; (void)aCompW;
#endif
    const unsigned otOff16 = (unsigned)(ot * LPG + q) * 16u;
    const unsigned tokM16 = (unsigned)((it * (JNOUT + 1) + ot) * LPG + q) * 16u;
    (void)aCur; (void)aDiag; (void)tokM16; (void)otOff16; (void)aLeft; (void)aDown;
    if (t == 0 && a == 0) {
      if (wv == 0) med_slow_supercell(desc, grec, P.nChunks, ldsb, aDiag, aLeft, aDown, aCur, aCompW, it, ot, q,
                                      P.seedOff, active && i == 0 && o == 0, active, aB, accBase, negLL, tbColOff, jdet);
    } else {
#if JTB
      unsigned char *tbCol = tbL + c * JTBS;
#endif
/*@BODY@*/
    }
    med_wave_sync();
#if JMODE == 2 && JFLAT
    // every state of this step's supercells is final: the usage of the transitions that apply to them (origin supercell included)
/*@FLAT@*/
    med_wave_sync();   // bvec is rewritten below
#endif
    // everything loaded from global memory in this step is consumed here, BEFORE the step's global stores are issued:
    // vmcnt counts loads and stores in order on gfx9, so a wait placed after the stores would wait for them too
    if (wantTok) {
#pragma unroll
      for (int k = 0; k < JTOKN; ++k) {
        const int j = tid + k * NT;
        if (j < W + C - 1) tokWin[((kw + 1) & 1) * (W + C) + j] = tokPre[k];
#if JENV
        if (j < W + C - 1) { envSW[((kw + 1) & 1) * (W + C) + j] = envSPre[k]; envEW[((kw + 1) & 1) * (W + C) + j] = envEPre[k]; }
#endif
      }
    }
#if JMODE == 2
#pragma unroll
#if JBDIST == 2
    for (int k = 0; k < JBV; ++k) { const int j = k * LPG + q; if (j < S) bvec[c * Spad + j] = JFLAT ? bnext[k] + negLL : bnext[k]; }
#pragma unroll
    for (int k = 0; k < JBV; ++k) bnext[k] = bpre[k];
#else
    for (int k = 0; k < JBV; ++k) { const int j = k * LPG + q; if (j < S && !(JDBG & 32)) bvec[c * Spad + j] = JFLAT ? bpre[k] + negLL : bpre[k]; }
#endif
#endif
#if JMAT == 2
    if (wantHalo) {
#pragma unroll
#if JCR
      for (int k = 0; k < JNHR; ++k) if (tid + k * NT < JNH) shortOf(slotCur, 0)[tid + k * NT] = hvr[k];
#else
      for (int k = 0; k < JNHR; ++k) if (tid + k * NT < JNH) ring(slotCur, 0)[myHS[k]] = hvr[k];
#endif
    }
#elif JCR
    if (wantHalo) {
      double *hd = shortOf(slotCur, 0);
#pragma unroll
      for (int k = 0; k < JNHR; ++k) if (tid + k * NT < JNH) hd[tid + k * NT] = hvc[k];
    }
#else
    if (wantHalo) {
      double *hd = ring(slotCur, 0);
#pragma unroll
      for (int k = 0; k < JHALO; ++k) {
        const int j = tid + k * NT;
#if JHALOT > 0
        if (j < S) hd[j] = haloBuf[(t - t0) * S + j];
#else
        if (j < S) hd[j] = hv[k];
#endif
      }
    }
#endif
    const double *cur = (const double *)(ldsb + aCur);
    if (active) {
#if JMAT == 1
      {
        double *dstp = cellPtr(i, o);
        med_copy_out(dstp, cur, q);
      }
#elif JMAT == 0
      if (c == C - 1) med_copy_out(haloOut + (long long)o * S, cur, q);
#else
      if (c == C - 1 && a + 1 < NA)
        for (int k = q; k < JNH; k += LPG) haloOut[(long long)o * JNHP + k] = cur[JHSTATE(k)];
#endif
#if JTB
      {   // the supercell's traceback bytes, 16 (or 4) per lane and store, the lanes of the column side by side
        unsigned char *dstb = tbPair + ((long long)o * I + i) * JSB;
        const unsigned char *srcb = tbL + c * JTBS;
#if JSB % 16 == 0
        for (int j = q * 16; j < JSB; j += LPG * 16) *(u32x4 *)(dstb + j) = *(const u32x4 *)(srcb + j);
#else
        for (int j = q * 4; j < JSB; j += LPG * 4) *(unsigned int *)(dstb + j) = *(const unsigned int *)(srcb + j);
#endif
      }
#endif
      if (i == inLen && o == outLen && q == 0 && A.loglike) A.loglike[pairIdx] = cur[JENDNODE];
    }
#if JNBSYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // this step's LDS writes are done before the counter says so
    if (lane == 0) stepDone[wv] = t - t0 + 1;
    if (((t + 1) % W) == 0 || t + 1 == t1) med_block_sync();      // end of a token window (the next step rewrites the other buffer) / of the tile
#else
    med_block_sync();
#endif
    slotCur = (slotCur + 1) % NS;
  }
#if JMAT == 2
  // what the next block of this strip starts from: the ring slots of the last NS - 1 steps, every column
  if (t1 < T)
    for (int dt = 1; dt < NS; ++dt) {
      const int slot = (((t1 - dt) % NS) + NS) % NS;
#if JCR
      if (dt == 1) for (int idx = tid; idx < C * S; idx += NT) { const int cc = idx / S, j = idx - cc * S; bnd[idx] = curOf(cc + 1)[j]; }      // the cells of step t1 - 1
      for (int idx = tid; idx < C * JNHP; idx += NT) { const int cc = idx / JNHP, k = idx - cc * JNHP; bnd[(long long)C * S + (long long)(dt - 1) * C * JNHP + idx] = shortOf(slot, cc + 1)[k]; }
#else
      for (int idx = tid; idx < C * S; idx += NT) {
        const int cc = idx / S, j = idx - cc * S;
        bnd[(long long)(dt - 1) * C * S + idx] = ring(slot, cc + 1)[j];
      }
#endif
    }
#endif
#if JMODE == 2
  // usage summed in registers over the tile's steps (records held in VGPRs) -> the workgroup's LDS accumulators
#if JNALL > 0
  // ... which, for a flat program, is a table of its own: one entry per transition, laid over the start of the LDS now that the ring,
  // the records and the Backward supercells are dead (the loop-time table, JNLOOP entries for the token-selected usage records, lies behind it)
  __syncthreads();
  for (int j = tid; j < JNALL; j += NT) lds[j] = 0.0;
  __syncthreads();
  const unsigned accBase2 = 0u;
#else
  const unsigned accBase2 = accBase;
#endif
  (void)accBase2;
/*@POST@*/
  __syncthreads();
  // flush the workgroup's counts: one fp64 atomic per transition that was used in this tile
#if JNALL > 0
  for (int e = tid; e < JNTRANS + JNLOOP; e += NT) {
    const bool loop = e >= JNTRANS;
    const double *tab = loop ? accL : lds;
    const int j = loop ? e - JNTRANS : e, tr = loop ? P.accMap[j] : e;
#else
  for (int e = tid; e < JNTRANS; e += NT) {
    const double *tab = accL; const int j = e, tr = e;
#endif
    if (jdet) {      // 2^-44 in the tile -> 2^-36 in global memory, rounded
      const unsigned long long u = ((const unsigned long long *)tab)[j];
      if (u) (void)__hip_atomic_fetch_add((unsigned long long *)A.counts + tr, u >= (1ull << 62) ? (1ull << 62) : (u + 128ull) >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a saturated tile stays saturated
      continue;
    }
    const double x = tab[j];
    if (x != 0.0) (void)__hip_atomic_fetch_add(A.counts + tr, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}
)MBJIT";
}  // namespace mb

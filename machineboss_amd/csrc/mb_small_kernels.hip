// mb_small_kernels.hip -- ahead-of-time kernels of the small-machine family (mb_small.cpp): conversion of its tile-major
// matrices to the reference's layout, and the Viterbi traceback over one-byte-per-cell pointers.
#include <algorithm>
#include <cstdlib>

#include "mb_internal.h"
#include "mb_small.h"

namespace mb {

// tile-major (strip, step, chunk, lane) -> IdentityIndexMapper layout ((outPos * (inLen+1)) + inPos) * nStates + state
// (src/dpmatrix.h:34-44,90-96).  `reversed`: the matrix was filled by the Backward sweep, which runs in the reversed frame.
__global__ __launch_bounds__(256) void k_small_unpack(const double *__restrict__ pool, int S, int inLen, int outLen, int reversed,
                                                      double *__restrict__ cells, const int *__restrict__ envStart,
                                                      const int *__restrict__ envEnd) {
  const long long I = inLen + 1, n = I * (outLen + 1) * S;
  const int CBD = small_chunk_bytes(S) / 8, NCH = small_chunks(S), Te = small_steps(outLen);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const long long sc = idx / S;
    const int s = (int)(idx - sc * S);
    const int o = (int)(sc / I), i = (int)(sc - (long long)o * I);
    if (envStart && (i < envStart[o] || i >= envEnd[o])) continue;   // outside the envelope: the caller's -inf stays (tiles there may not have run)
    // the reversed frame keeps its padding columns in front (mb_small.cpp): frame column inLen - i sits at NA*64 - 1 - i
    const int fi = reversed ? small_strips(inLen) * 64 - 1 - i : i, fo = reversed ? outLen - o : o;
    const int a = fi >> 6, c = fi & 63, t = fo + c;
    cells[idx] = pool[((((long long)a * Te + t) * NCH + s / CBD) * 64 + c) * CBD + (s % CBD)];
  }
}

int launch_small_unpack(const double *d_pool, int S, int inLen, int outLen, bool reversed, double *d_cells, const int *d_envStart,
                        const int *d_envEnd, hipStream_t st) {
  const long long n = (long long)(inLen + 1) * (outLen + 1) * S;
  hipLaunchKernelGGL(k_small_unpack, dim3((unsigned)std::min<long long>((n + 255) / 256, 16384)), dim3(256), 0, st, d_pool, S, inLen,
                     outLen, reversed ? 1 : 0, d_cells, d_envStart, d_envEnd);
  return hip_ok(hipGetLastError(), "unpack launch") ? 0 : 1;
}

// DPMatrix::traceBack with selectMaxTrans (src/dpmatrix.defs.h:82-110,171-174) over the traceback bytes the SM_TB sweep
// stored: byte (i,o,s) = index of the first maximal candidate of that cell in the reference's enumeration order (match,
// input-only, output-only, silent; ascending source state, then insertion order), which is the choice std::max_element
// makes there.  One LANE per pair: a step is a byte and two table look-ups, no candidate is re-evaluated.
// Edge ids are written backwards from the end of the pair's slot; pathLen[p] = number of transitions (-1: end cell is
// -inf, -2: slot too small).
struct SmTbTables {
  const int *decOff; const uint32_t *dec; const int *eid;
  long long off0, off1, off2, off3;
  int S, nIn, nOut, tbStride;
};

// The walk is a chain of dependent look-ups, so what matters is the number of trips to memory per path step:
//   * the traceback bytes of a supercell are one to four dwords; the lane holds the dwords of the supercell it stands on, so
//     silent moves (same supercell) cost no memory access;
//   * on arrival at a supercell the dwords of its three possible predecessors (i-1,o-1), (i-1,o), (i,o-1) and the two
//     tokens in[i-2], out[o-2] are requested at once; the move into one of them happens after the silent chain in
//     between, so an emitting move costs at most ONE memory round trip;
//   * the decode and edge-id tables sit in LDS.
template <int NW>
__global__ __launch_bounds__(64) void k_small_traceback(SmTbTables T, int nDec, int nEid, const PairDesc *__restrict__ pairs, long long nPairs,
                                                        const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                        const unsigned char *__restrict__ tb, const SmAux *__restrict__ aux,
                                                        const double *__restrict__ ll, const long long *__restrict__ slotOff,
                                                        uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  extern __shared__ int tbl[];
  int *lDecOff = tbl;                          // [S + 1]
  uint32_t *lDec = (uint32_t *)(tbl + T.S + 1);   // [nDec]
  int *lEid = (int *)(lDec + nDec);            // [nEid] (0: the table stays in global memory)
  for (int k = threadIdx.x; k <= T.S; k += 64) lDecOff[k] = T.decOff[k];
  for (int k = threadIdx.x; k < nDec; k += 64) lDec[k] = T.dec[k];
  for (int k = threadIdx.x; k < nEid; k += 64) lEid[k] = T.eid[k];
  __syncthreads();
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPairs) return;
  if (!(ll[p] > -INFINITY)) { pathLen[p] = -1; return; }
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const int Te = small_steps(outLen);
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const uint32_t *words = (const uint32_t *)(tb + aux[p].tb);
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  struct Cell { uint32_t w[NW]; };
  auto load = [&](int ci, int co) -> Cell {   // clamped: cells outside the lattice are never moved into
    Cell c;
    const int x = max(ci, 0), y = max(co, 0);
    const uint32_t *q = words + (((long long)(x >> 6) * Te + (y + (x & 63))) * 64 + (x & 63)) * NW;
#pragma unroll
    for (int k = 0; k < NW; ++k) c.w[k] = q[k];
    return c;
  };
  int i = inLen, o = outLen, s = T.S - 1;
  int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;
  Cell cur = load(i, o), cd = load(i - 1, o - 1), cl = load(i - 1, o), cu = load(i, o - 1);
  int itP = i > 1 ? in[i - 2] : 0, otP = o > 1 ? out[o - 2] : 0;
  long long n = 0;
  while (i > 0 || o > 0 || s != 0) {
    uint32_t wv = cur.w[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) wv = (s >> 2) == k ? cur.w[k] : wv;
    const unsigned kc = (wv >> (8 * (s & 3))) & 255u;
    const uint32_t d = lDec[lDecOff[s] + (int)kc];
    const int kind = (int)(d & 255u), src = (int)((d >> 8) & 255u), tab = (int)(d >> 16);
    long long e;
    if (kind == 0) e = T.off0 + (long long)tab * (T.nIn + 1) * (T.nOut + 1) + (long long)it * (T.nOut + 1) + ot;
    else if (kind == 1) e = T.off1 + (long long)tab * (T.nIn + 1) + it;
    else if (kind == 2) e = T.off2 + (long long)tab * (T.nOut + 1) + ot;
    else e = T.off3 + tab;
    if (n >= cap) { pathLen[p] = -2; return; }
    pathBuf[slot0 + cap - 1 - n] = (uint32_t)(nEid ? lEid[e] : T.eid[e]);
    ++n;
    s = src;
    if (kind != 3) {
      if (kind == 0) { cur = cd; --i; --o; it = itP; ot = otP; }
      else if (kind == 1) { cur = cl; --i; it = itP; }
      else { cur = cu; --o; ot = otP; }
      cd = load(i - 1, o - 1); cl = load(i - 1, o); cu = load(i, o - 1);
      itP = i > 1 ? in[i - 2] : 0; otP = o > 1 ? out[o - 2] : 0;
    }
  }
  pathLen[p] = n;
}

// The same walk with ONE WAVEFRONT per pair and a WINDOW of traceback words in LDS.  A move goes from (strip step t, lane c)
// to (t - 1 or t - 2, c or c - 1), so a block of WT steps x WC lanes whose corner is the current cell holds the next
// min(WT / 2, WC) moves at least: the wavefront fetches it with coalesced loads (WC x NW dwords = one or two cache lines
// per step row), together with the input / output tokens of its rows and columns, and then walks inside LDS -- a step
// costs three dependent LDS look-ups (traceback word, decode entry, edge id) instead of a trip to HBM.  Everything
// about the position is wavefront-uniform (readfirstlane); lane 0 writes the path.
template <int NW, int WT, int WC, bool EIDLDS>
__global__ __launch_bounds__(256) void k_small_traceback_wave(SmTbTables T, int nDec, int nEid, const PairDesc *__restrict__ pairs, long long nPairs,
                                                             const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                             const unsigned char *__restrict__ tb, const SmAux *__restrict__ aux,
                                                             const double *__restrict__ ll, const long long *__restrict__ slotOff,
                                                             uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  extern __shared__ int tbl[];
  int *lDecOff = tbl;                          // [S + 1]
  // decode entry of (state, candidate index), prepared for the walk: {source state | consumes input << 8 | consumes output << 9,
  // first entry of the candidate's table, entries per input token | entries per output token << 16} -- the entry of the
  // edge-id table is base + inTok * mulI + outTok * mulO
  uint32_t *lDec = (uint32_t *)(tbl + T.S + 1);   // [nDec][3]
  int *lEid = (int *)(lDec + 3 * nDec);        // [nEid] (0: the table stays in global memory)
  for (int k = threadIdx.x; k <= T.S; k += 256) lDecOff[k] = T.decOff[k];
  for (int k = threadIdx.x; k < nDec; k += 256) {
    const uint32_t d = T.dec[k];
    const int kind = (int)(d & 255u), tab = (int)(d >> 16);
    const uint32_t src = (d >> 8) & 255u;
    long long base; uint32_t mulI = 0, mulO = 0;
    if (kind == 0) { base = T.off0 + (long long)tab * (T.nIn + 1) * (T.nOut + 1); mulI = (uint32_t)(T.nOut + 1); mulO = 1; }
    else if (kind == 1) { base = T.off1 + (long long)tab * (T.nIn + 1); mulI = 1; }
    else if (kind == 2) { base = T.off2 + (long long)tab * (T.nOut + 1); mulO = 1; }
    else base = T.off3 + tab;
    lDec[3 * k] = src | ((kind == 0 || kind == 1) ? 256u : 0u) | ((kind == 0 || kind == 2) ? 512u : 0u);
    lDec[3 * k + 1] = (uint32_t)base;
    lDec[3 * k + 2] = mulI | (mulO << 16);
  }
  for (int k = threadIdx.x; k < nEid; k += 256) lEid[k] = T.eid[k];
  __syncthreads();
  constexpr int WIN = WT * WC * NW, PERW = WIN + WC + WT + 64;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint32_t *win = (uint32_t *)(lEid + nEid) + wv * PERW;   // [WT][WC][NW]: row dt = tTop - t, column dc = c - (cRight - WC + 1)
  int *tokI = (int *)(win + WIN);              // tokI[k] = in[iA - 1 - k]: the token consumed on leaving column iA - k
  int *tokO = tokI + WC;                       // tokO[k] = out[oA - 1 - k]
  // Edge ids go through a 64-entry block in LDS and reach the path 64 at a time (one coalesced store): on gfx9 a store
  // counts in vmcnt like a load, so a per-step store would make every later wait for a load -- the next window -- also a
  // wait for the store's acknowledgement, and with the edge-id table in LDS the walk then has no vector-memory wait at all.
  uint32_t *pbuf = (uint32_t *)(tokO + WT);
  const long long p = (long long)blockIdx.x * 4 + wv;
  if (p >= nPairs) return;
  if (!(ll[p] > -INFINITY)) { if (lane == 0) pathLen[p] = -1; return; }
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const int Te = small_steps(outLen);
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const uint32_t *words = (const uint32_t *)(tb + aux[p].tb);
  const long long slot0 = slotOff[p];
  const int cap = (int)min(slotOff[p + 1] - slot0, 0x7fffffffll);   // (a path has fewer than 2^31 transitions)
  int i = inLen, o = outLen, s = T.S - 1;
  int a = -1, tTop = 0, cRight = 0, iA = 0, oA = 0;   // window: strip, corner (step, lane), and the cell (iA, oA) at the corner
  int n = 0;
  // the block of traceback words below the current cell, if the cell has left the one in LDS
  auto ensure_window = [&]() {
    const int c = i & 63, t = o + c;
    if ((i >> 6) == a && t > tTop - WT && c > cRight - WC) return;
    a = i >> 6; tTop = t; cRight = c; iA = i; oA = o;
    __builtin_amdgcn_wave_barrier();
    const uint32_t *base = words + ((long long)a * Te * 64) * NW;
    // 16 bytes per lane and load (a window row is WC x NW = 32 ... 64 contiguous dwords, 4-byte aligned: global_load_dwordx4 takes
    // that), 8 loads in flight: a 16 KB window is two round trips to memory instead of eight (round 4; the one-tape code walker's
    // window taught it).  Entries left of lane 0 or above step 0 are never looked up, so whatever a vector brings
    // for them stays; only a vector that would START before the pair's first byte (strip 0, step 0) is read element by element.
    constexpr int PER = WIN / 256, BLK = PER < 8 ? PER : 8;       // 16-byte loads per lane
    static_assert(WIN % 256 == 0 && (WC * NW) % 4 == 0 && PER % BLK == 0, "window size");
    typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
    const int c0 = cRight - WC + 1;                               // first lane of the window's rows
    const int tiv = inLen > 0 ? in[max(iA - 1 - lane, 0)] : 0, tov = outLen > 0 ? out[max(oA - 1 - lane, 0)] : 0;   // requested with the block (clamped indices)
#pragma unroll 1
    for (int k0 = 0; k0 < PER; k0 += BLK) {
      u32x4a4 v[BLK];
#pragma unroll
      for (int k = 0; k < BLK; ++k) {
        const int idx = ((k0 + k) * 64 + lane) * 4, dt = idx / (WC * NW), rem = idx - dt * (WC * NW);
        const int tt = tTop - dt;
        const long long off = ((long long)tt * 64 + c0) * NW + rem;
        if (tt >= 0 && (off >= 0 || a > 0)) v[k] = *(const u32x4a4 *)(base + off);
        else {
          u32x4a4 z;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const bool ok = tt >= 0 && c0 + (rem + j) / NW >= 0; const uint32_t x = base[ok ? off + j : 0]; z[j] = ok ? x : 0u; }
          v[k] = z;
        }
      }
#pragma unroll
      for (int k = 0; k < BLK; ++k) *(u32x4a4 *)(win + ((k0 + k) * 64 + lane) * 4) = v[k];
    }
    static_assert(WC <= 64 && WT <= 64, "token windows are one load per lane");
    if (lane < WC) tokI[lane] = (iA - 1 - lane >= 0) ? tiv : 0;
    if (lane < WT) tokO[lane] = (oA - 1 - lane >= 0) ? tov : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto win_at = [&]() -> uint32_t {
    const int c = i & 63, t = o + c;
    return win[((tTop - t) * WC + (c - (cRight - WC + 1))) * NW + (s >> 2)];
  };
  bool overflow = false;
  auto emit = [&](uint32_t eid) {              // (every lane holds the same value; no branch before the LDS write, so that the look-up of `eid` is not sunk behind one)
    overflow = overflow || n >= cap;
    pbuf[n & 63] = eid;
    ++n;
    if ((n & 63) == 0 && !overflow) {          // a full block: entries n-64 .. n-1, stored backwards from the end of the slot
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = pbuf[lane];
      __builtin_amdgcn_wave_barrier();
    }
  };
  // A step has a CHAIN the next step waits for -- traceback word -> decode entry -> (source state, which tapes move) -- and
  // a TAIL nothing waits for -- table entry + tokens -> edge id -> path.  The tail of step k is issued at the top of step
  // k + 1, so its LDS round trips overlap the chain's: two LDS latencies per step instead of four.  Position and state are
  // wavefront-uniform and kept in scalar registers (readfirstlane of the two look-ups of the chain).
  if (i > 0 || o > 0 || s != 0) {
    int pk3, pti, pto;                         // the previous step's decode entry and token-window indices
    {
      ensure_window();
      const uint32_t wv32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)win_at());
      pk3 = 3 * (lDecOff[s] + (int)((wv32 >> (8 * (s & 3))) & 255u));
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lDec[pk3]);
      pti = iA - i; pto = oA - o;              // tokens in[i - 1], out[o - 1]
      s = (int)(w0 & 255u); i -= (int)((w0 >> 8) & 1u); o -= (int)((w0 >> 9) & 1u);
    }
    while ((i > 0 || o > 0 || s != 0) && !overflow) {
      const uint32_t w1 = lDec[pk3 + 1], w2 = lDec[pk3 + 2], ti = (uint32_t)tokI[pti], to = (uint32_t)tokO[pto];   // tail of the previous step (LDS executes in order: a window fetched below does not disturb these reads)
      ensure_window();
      const uint32_t wvv = win_at();
      const int dOff = lDecOff[s];
      const uint32_t e = w1 + ti * (w2 & 0xffffu) + to * (w2 >> 16);
      const uint32_t wv32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wvv);
      const int k3 = 3 * (dOff + (int)((wv32 >> (8 * (s & 3))) & 255u));
      const uint32_t w0v = lDec[k3];
      const uint32_t eid = (uint32_t)(EIDLDS ? lEid[e] : T.eid[e]);
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w0v);
      emit(eid);
      pk3 = k3; pti = iA - i; pto = oA - o;
      s = (int)(w0 & 255u); i -= (int)((w0 >> 8) & 1u); o -= (int)((w0 >> 9) & 1u);
    }
    if (!overflow) {                           // the last step's tail
      const uint32_t w1 = lDec[pk3 + 1], w2 = lDec[pk3 + 2], ti = (uint32_t)tokI[pti], to = (uint32_t)tokO[pto];
      const uint32_t e = w1 + ti * (w2 & 0xffffu) + to * (w2 >> 16);
      emit((uint32_t)(EIDLDS ? lEid[e] : T.eid[e]));
    }
  }
  if (overflow) { if (lane == 0) pathLen[p] = -2; return; }
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int done = n & ~63;
    if (done + lane < n) pathBuf[slot0 + cap - 1 - (done + lane)] = pbuf[lane];
  }
  if (lane == 0) pathLen[p] = n;
}

template <int NW>
static void launch_tb_wave(const SmTbTables &T, int nDec, int nEid, size_t tblBytes, const PairDesc *d_pairs, long long nPairs, const int *d_in,
                           const int *d_out, const unsigned char *d_tb, const SmAux *d_aux, const double *d_ll, const long long *d_slotOff,
                           uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st) {
  constexpr int WT = NW <= 2 ? 64 : 32, WC = NW <= 2 ? 32 : 16;
  const size_t lds = tblBytes + 4 * (size_t)(WT * WC * NW + WC + WT + 64) * 4;
  static bool attr = false;   // (one flag per instantiation: the table + four windows may exceed the default 64 KB)
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)&k_small_traceback_wave<NW, WT, WC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)&k_small_traceback_wave<NW, WT, WC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)((nPairs + 3) / 4)), block(256);
  if (nEid) hipLaunchKernelGGL((k_small_traceback_wave<NW, WT, WC, true>), grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen);
  else hipLaunchKernelGGL((k_small_traceback_wave<NW, WT, WC, false>), grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen);
}

int launch_small_traceback(const SmallProgram &P, const PairDesc *d_pairs, long long nPairs, const int *d_in, const int *d_out,
                           const unsigned char *d_tb, const SmAux *d_aux, const double *d_ll, const long long *d_slotOff,
                           uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st) {
  if (nPairs == 0) return 0;
  SmTbTables T;
  T.decOff = P.d_decOff; T.dec = P.d_dec; T.eid = P.d_eid;
  T.off0 = P.off[0]; T.off1 = P.off[1]; T.off2 = P.off[2]; T.off3 = P.off[3];
  T.S = P.S; T.nIn = P.nIn; T.nOut = P.nOut; T.tbStride = small_tb_stride(P.S);
  const int nDec = (int)P.dec.size(), nEid = P.nEntries <= 12288 ? (int)P.nEntries : 0;
  const size_t lds = (size_t)(P.S + 1 + nDec + nEid) * 4, ldsWave = (size_t)(P.S + 1 + 3 * nDec + nEid) * 4;
  // one wavefront per pair unless the batch is so large that one lane per pair already fills the chip
  static const int waveMax = []() { const char *e = opt_env("MB_SMALL_TRACEBACK_WAVE_MAX_PAIRS"); return e && *e ? atoi(e) : 262144; }();
  if (nPairs <= waveMax) {
    switch (T.tbStride / 4) {
      case 1: launch_tb_wave<1>(T, nDec, nEid, ldsWave, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen, st); break;
      case 2: launch_tb_wave<2>(T, nDec, nEid, ldsWave, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen, st); break;
      case 3: launch_tb_wave<3>(T, nDec, nEid, ldsWave, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen, st); break;
      default: launch_tb_wave<4>(T, nDec, nEid, ldsWave, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen, st); break;
    }
    return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
  }
  const dim3 grid((unsigned)((nPairs + 63) / 64)), block(64);
  switch (T.tbStride / 4) {
    case 1: hipLaunchKernelGGL(k_small_traceback<1>, grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen); break;
    case 2: hipLaunchKernelGGL(k_small_traceback<2>, grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen); break;
    case 3: hipLaunchKernelGGL(k_small_traceback<3>, grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen); break;
    default: hipLaunchKernelGGL(k_small_traceback<4>, grid, block, lds, st, T, nDec, nEid, d_pairs, nPairs, d_in, d_out, d_tb, d_aux, d_ll, d_slotOff, d_pathBuf, d_pathLen); break;
  }
  return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
}

}  // namespace mb

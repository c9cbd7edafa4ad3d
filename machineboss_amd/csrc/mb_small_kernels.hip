// mb_small_kernels.hip -- ahead-of-time kernels of the small-machine family (mb_small.cpp): conversion of its tile-major
// matrices to the reference's layout, and the Viterbi traceback over one-byte-per-cell pointers.
#include <algorithm>

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

int launch_small_traceback(const SmallProgram &P, const PairDesc *d_pairs, long long nPairs, const int *d_in, const int *d_out,
                           const unsigned char *d_tb, const SmAux *d_aux, const double *d_ll, const long long *d_slotOff,
                           uint32_t *d_pathBuf, long long *d_pathLen, hipStream_t st) {
  if (nPairs == 0) return 0;
  SmTbTables T;
  T.decOff = P.d_decOff; T.dec = P.d_dec; T.eid = P.d_eid;
  T.off0 = P.off[0]; T.off1 = P.off[1]; T.off2 = P.off[2]; T.off3 = P.off[3];
  T.S = P.S; T.nIn = P.nIn; T.nOut = P.nOut; T.tbStride = small_tb_stride(P.S);
  const int nDec = (int)P.dec.size(), nEid = P.nEntries <= 12288 ? (int)P.nEntries : 0;
  const size_t lds = (size_t)(P.S + 1 + nDec + nEid) * 4;
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

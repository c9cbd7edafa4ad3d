// mb_generic.hip -- generic HIP kernels: correct for ANY advancing machine, no LDS tiling.
//
// One workgroup per sequence pair; the lattice is swept along anti-diagonals; inside one anti-diagonal the
// states of all its supercells are finalised silent-level by silent-level (one __syncthreads() per level).
// The matrix lives in HBM in the reference's layout (src/dpmatrix.h:90-96), so this family is also what backs
// mb_fill(), Backward matrices, the posterior-count sweep and the Viterbi traceback for every machine size.
// The fast families (mb_medium.hip: lanes = states, tiled; mb_wide.hip: one workgroup per one-tape sequence) are checked
// against it.
#include <algorithm>

#include "mb_internal.h"
#include "mb_device_math.h"

namespace mb {

static int env_int_g(const char *name, int dflt) { const char *v = opt_env(name); return v && *v ? atoi(v) : dflt; }

// ---- DPMatrix::accumulate over `incoming` (src/dpmatrix.h:101-115) -----------------------------------------
template <int MODE>
__device__ __forceinline__ double fold_in(const DevMachine &m, double acc, int d, int it, int ot,
                                          const double *srcCell, bool silent) {
  const int row = (d * (m.nIn + 1) + it) * (m.nOut + 1) + ot;
  const int a0 = m.inOff[row], a1 = m.inOff[row + 1];
  for (int a = a0; a < a1; ++a) {
    const int s = (int)m.inSrc[a];
    if (silent && s >= d) continue;  // silent self-loop on state 0: reads a not-yet-written (-inf) cell in the reference
    const double v = srcCell[s] + m.inW[a];
    acc = (MODE == MB_VITERBI) ? dmax(acc, v) : lse2_exact(acc, v);
  }
  return acc;
}

template <int MODE>
__device__ __forceinline__ double fold_out(const DevMachine &m, double acc, int s, int it, int ot,
                                           const double *dstCell, bool silent) {
  const int row = (s * (m.nIn + 1) + it) * (m.nOut + 1) + ot;
  const int a0 = m.outOff[row], a1 = m.outOff[row + 1];
  for (int a = a0; a < a1; ++a) {
    const int d = (int)m.outDst[a];
    if (silent && d <= s) continue;
    acc = lse2_exact(acc, dstCell[d] + m.outW[a]);
  }
  return acc;
}

// MappedForwardMatrix::fill (src/forward.defs.h:23-49) / ViterbiMatrix::fill (src/viterbi.cpp:18-43)
template <int MODE>
__global__ __launch_bounds__(1024) void k_generic_fill_fwd(DevMachine m, const PairDesc *__restrict__ pairs,
                                                           const int *__restrict__ inTok,
                                                           const int *__restrict__ outTok,
                                                           double *pool, int startState,
                                                           const int *__restrict__ envStart, const int *__restrict__ envEnd) {
  const PairDesc pd = pairs[blockIdx.x];
  const int inLen = pd.inLen, outLen = pd.outLen, S = m.S;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  double *cells = pool + pd.cellBase;
  for (int diag = 0; diag <= inLen + outLen; ++diag) {
    const int iLo = diag > outLen ? diag - outLen : 0;
    const int iHi = diag < inLen ? diag : inLen;
    const int n = iHi - iLo + 1;
    for (int lev = 0; lev < m.nLevF; ++lev) {
      const int l0 = m.levFOff[lev], ns = m.levFOff[lev + 1] - l0;
      for (int idx = threadIdx.x; idx < n * ns; idx += blockDim.x) {
        const int k = idx / ns, j = idx - k * ns;
        const int i = iLo + k, o = diag - i;
        const int d = m.levFState[l0 + j];
        // cells outside the envelope keep the -inf the pool was filled with (src/dpmatrix.defs.h:36, dpmatrix.h:142-144)
        if (pd.envBase >= 0 && (i < envStart[pd.envBase + o] || i >= envEnd[pd.envBase + o])) continue;
        const int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;
        double *cur = cells + ((long long)o * I + i) * S;
        double acc = (i || o || d != startState) ? -INFINITY : 0.0;
        if (i && o) acc = fold_in<MODE>(m, acc, d, it, ot, cur - (I + 1) * S, false);
        if (i) acc = fold_in<MODE>(m, acc, d, it, 0, cur - S, false);
        if (o) acc = fold_in<MODE>(m, acc, d, 0, ot, cur - I * S, false);
        acc = fold_in<MODE>(m, acc, d, 0, 0, cur, true);
        cur[d] = acc;
      }
      __syncthreads();
    }
  }
}

// BackwardMatrix::fill (src/backward.cpp:18-46)
__global__ __launch_bounds__(1024) void k_generic_fill_bwd(DevMachine m, const PairDesc *__restrict__ pairs,
                                                           const int *__restrict__ inTok,
                                                           const int *__restrict__ outTok,
                                                           double *pool,
                                                           const int *__restrict__ envStart, const int *__restrict__ envEnd) {
  const PairDesc pd = pairs[blockIdx.x];
  const int inLen = pd.inLen, outLen = pd.outLen, S = m.S;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  double *cells = pool + pd.cellBase;
  for (int diag = inLen + outLen; diag >= 0; --diag) {
    const int iLo = diag > outLen ? diag - outLen : 0;
    const int iHi = diag < inLen ? diag : inLen;
    const int n = iHi - iLo + 1;
    for (int lev = 0; lev < m.nLevB; ++lev) {
      const int l0 = m.levBOff[lev], ns = m.levBOff[lev + 1] - l0;
      for (int idx = threadIdx.x; idx < n * ns; idx += blockDim.x) {
        const int k = idx / ns, j = idx - k * ns;
        const int i = iLo + k, o = diag - i;
        const int s = m.levBState[l0 + j];
        if (pd.envBase >= 0 && (i < envStart[pd.envBase + o] || i >= envEnd[pd.envBase + o])) continue;
        const bool endIn = (i == inLen), endOut = (o == outLen);
        const int it = endIn ? 0 : in[i], ot = endOut ? 0 : out[o];
        double *cur = cells + ((long long)o * I + i) * S;
        double acc = (endIn && endOut && s == S - 1) ? 0.0 : -INFINITY;
        if (!endIn && !endOut) acc = fold_out<MB_FORWARD>(m, acc, s, it, ot, cur + (I + 1) * S, false);
        if (!endIn) acc = fold_out<MB_FORWARD>(m, acc, s, it, 0, cur + S, false);
        if (!endOut) acc = fold_out<MB_FORWARD>(m, acc, s, 0, ot, cur + I * S, false);
        acc = fold_out<MB_FORWARD>(m, acc, s, 0, 0, cur, true);
        cur[s] = acc;
      }
      __syncthreads();
    }
  }
}

// loglike[p] = cell(inLen,outLen,endState) (forward / Viterbi, src/forward.defs.h:51-55, viterbi.cpp:45-47)
// or cell(0,0,startState) (backward, src/backward.cpp:48-50)
__global__ void k_gather_loglike(const PairDesc *__restrict__ pairs, long long nPairs, const double *__restrict__ pool,
                                 int S, int backward, double *__restrict__ loglike) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPairs) return;
  const PairDesc pd = pairs[p];
  const long long n = (long long)(pd.inLen + 1) * (pd.outLen + 1) * S;
  loglike[p] = backward ? pool[pd.cellBase] : pool[pd.cellBase + n - 1];
}

// BackwardMatrix::getCounts (src/backward.cpp:58-87): for every cell and every outgoing edge
//   count[src][ti] += exp( F(i,o,src) - B(0,0,start) + (B(i',o',dst) + logW) )
// The sweep has no dependencies, so it is a flat grid over (pair, supercell, state); per-workgroup partial
// counts are kept in LDS when the transition table is small and flushed once with fp64 atomics.
#define MB_COUNTS_LDS_MAX 8192
__global__ __launch_bounds__(256) void k_generic_counts(DevMachine m, const PairDesc *__restrict__ pairs,
                                                        const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                        const double *__restrict__ fwdPool,
                                                        const double *__restrict__ bwdPool, long long nTrans,
                                                        int chunksPerPair, double *__restrict__ counts, int det) {
  // det (deterministic mode, mb_internal.h): both tables hold 64-bit fixed point at 2^-36 -- every term is below 1, integer adds commute
  __shared__ double lcount[MB_COUNTS_LDS_MAX];
  const bool useLds = nTrans <= MB_COUNTS_LDS_MAX;
  if (useLds) {
    for (int e = threadIdx.x; e < nTrans; e += blockDim.x) lcount[e] = 0.0;
    __syncthreads();
  }
  const int p = blockIdx.x / chunksPerPair, chunk = blockIdx.x % chunksPerPair;
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen, S = m.S;
  const long long I = inLen + 1, O = outLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const double *F = fwdPool + pd.cellBase, *B = bwdPool + pd.cellBase;
  const double ll = B[0];  // backward.logLike() (src/backward.cpp:66)
  if (ll > -INFINITY) {
    const long long nItems = I * O * S;
    for (long long idx = (long long)chunk * blockDim.x + threadIdx.x; idx < nItems;
         idx += (long long)chunksPerPair * blockDim.x) {
      const long long sc = idx / S;
      const int s = (int)(idx - sc * S);
      const int o = (int)(sc / I), i = (int)(sc - (long long)o * I);
      const double f = F[idx];
      if (!(f > -INFINITY)) continue;
      const double logOdds = f - ll;
      const bool endIn = (i == inLen), endOut = (o == outLen);
      const int it = endIn ? 0 : in[i], ot = endOut ? 0 : out[o];
      const double *cur = B + sc * S;
      for (int grp = 0; grp < 4; ++grp) {
        int kit, kot; const double *dc;
        if (grp == 0) { if (endIn || endOut) continue; kit = it; kot = ot; dc = cur + (I + 1) * S; }
        else if (grp == 1) { if (endIn) continue; kit = it; kot = 0; dc = cur + S; }
        else if (grp == 2) { if (endOut) continue; kit = 0; kot = ot; dc = cur + I * S; }
        else { kit = 0; kot = 0; dc = cur; }
        const int row = (s * (m.nIn + 1) + kit) * (m.nOut + 1) + kot;
        for (int a = m.outOff[row]; a < m.outOff[row + 1]; ++a) {
          const double tll = dc[m.outDst[a]] + m.outW[a];
          const double c = exp(logOdds + tll);
          if (c != 0.0) {
            double *tab = useLds ? lcount : counts;
            if (det) atomicAdd((unsigned long long *)tab + m.outEid[a], (unsigned long long)fmin(fmax(c * 68719476736.0 + 0.5, 0.0), 4611686018427387904.0));
            else atomicAdd(&tab[m.outEid[a]], c);
          }
        }
      }
    }
  }
  if (useLds) {
    __syncthreads();
    for (int e = threadIdx.x; e < nTrans; e += blockDim.x)
      if (det ? ((const unsigned long long *)lcount)[e] != 0ull : lcount[e] != 0.0) {
        if (det) atomicAdd((unsigned long long *)counts + e, ((const unsigned long long *)lcount)[e]);
        else atomicAdd(&counts[e], lcount[e]);
      }
  }
}

// DPMatrix::traceBack with selectMaxTrans (src/dpmatrix.defs.h:82-110,171-174).  One wavefront per pair: the
// candidate list of the current cell is enumerated in the reference order (match, in-only, out-only, silent;
// within a group the `incoming` order), lanes take candidates round-robin, and the FIRST maximum wins
// (max value, then lowest enumeration index) exactly as std::max_element does.
// Edge ids are written backwards from the end of the pair's slot; pathLen[p] = number of transitions
// (-1: end cell is -inf, -2: slot too small, -3: dead end).
__global__ __launch_bounds__(64) void k_traceback(DevMachine m, const PairDesc *__restrict__ pairs,
                                                  const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                  const double *__restrict__ pool,
                                                  const long long *__restrict__ slotOff,
                                                  uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen, S = m.S;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const double *cells = pool + pd.cellBase;
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  int i = inLen, o = outLen, s = S - 1;
  if (!(cells[((long long)o * I + i) * S + s] > -INFINITY)) { if (lane == 0) pathLen[p] = -1; return; }
  long long n = 0;
  uint32_t held = 0;
  int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;   // tokens one step ahead, see k_traceback_scan
  while (i > 0 || o > 0 || s != 0) {
    const int itP = i > 1 ? in[i - 2] : 0, otP = o > 1 ? out[o - 2] : 0;
    const double *cur = cells + ((long long)o * I + i) * S;
    double best = -INFINITY; int bestIdx = 0x7fffffff; int bestA = -1;
    int base = 0;
    for (int grp = 0; grp < 4; ++grp) {
      int kit, kot; const double *sc;
      if (grp == 0) { if (!(i && o)) continue; kit = it; kot = ot; sc = cur - (I + 1) * S; }
      else if (grp == 1) { if (!i) continue; kit = it; kot = 0; sc = cur - S; }
      else if (grp == 2) { if (!o) continue; kit = 0; kot = ot; sc = cur - I * S; }
      else { kit = 0; kot = 0; sc = cur; }
      const int row = (s * (m.nIn + 1) + kit) * (m.nOut + 1) + kot;
      const int a0 = m.inOff[row], a1 = m.inOff[row + 1];
      for (int a = a0 + lane; a < a1; a += 64) {
        const int src = (int)m.inSrc[a];
        // a silent self-loop on state 0 is a genuine candidate in the reference's traceback (it reads the final
        // cell value); keep it: v = cell + w can only tie or lose against the real predecessor unless w >= 0.
        const double v = sc[src] + m.inW[a];
        const int idx = base + (a - a0);
        if (bestA < 0 || v > best || (v == best && idx < bestIdx)) { best = v; bestIdx = idx; bestA = a; }
      }
      base += a1 - a0;
    }
    // wave-wide arg-max with first-index tie-break; lanes without a candidate carry bestA = -1
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bestIdx, off), oa = __shfl_xor(bestA, off);
      const bool take = oa >= 0 && (bestA < 0 || ov > best || (ov == best && oi < bestIdx));
      if (take) { best = ov; bestIdx = oi; bestA = oa; }
    }
    if (bestA < 0) { if (lane == 0) pathLen[p] = -3; return; }
    if (n >= cap) { if (lane == 0) pathLen[p] = -2; return; }
    const uint32_t eid = m.inEid[bestA];
    // lane (n mod 64) keeps the step's edge; 64 of them go out in one store.  (On gfx9 a store counts in vmcnt like a load:
    // a store per step would make the next step's wait for its cells also a wait for that store's acknowledgement.)
    if ((int)(n & 63) == lane) held = eid;
    ++n;
    if ((n & 63) == 0) pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = held;
    if (m.eInTok[eid]) { --i; it = itP; }
    if (m.eOutTok[eid]) { --o; ot = otP; }
    s = (int)m.inSrc[bestA];
  }
  if ((n & ~63ll) + lane < n) pathBuf[slot0 + cap - 1 - ((n & ~63ll) + lane)] = held;
  if (lane == 0) pathLen[p] = n;
}

// The same walk for machines whose incoming tables fit LDS (CSR offsets <= 8192 rows, <= 2048 transitions): four pairs
// per workgroup (one wavefront each) share an LDS copy of the tables, so a path step costs ONE trip to HBM (the cells the
// candidates point at) instead of three dependent ones (CSR offsets -> edges -> cells).  With few states per supercell
// (S <= 16) the 3 x 3 block of supercells around the position -- every cell the NEXT step can read -- is touched one
// step ahead, so that trip mostly ends in the XCD's L2.  Candidate order, tie-break and results are those of k_traceback.
// OFFLDS = false: the CSR offsets stay in global memory (large alphabets: psw2dna has 271 x 105 rows) while the edges -- what
// the inner loop reads -- still come from LDS: two trips to memory per step (offsets, cells) instead of six.
struct TbEdge { double w; uint32_t eid; uint16_t src; uint8_t hasIn, hasOut; };
template <bool OFFLDS>
__global__ __launch_bounds__(256) void k_traceback_lds(DevMachine m, const PairDesc *__restrict__ pairs, long long nPairs,
                                                       long long nTrans, const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                       const double *__restrict__ pool, const long long *__restrict__ slotOff,
                                                       uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  extern __shared__ unsigned char tb_raw[];
  const int nRows = m.S * m.K;
  int *lOff = (int *)tb_raw;                                                   // [nRows + 1]
  TbEdge *lEdge = (TbEdge *)(tb_raw + (OFFLDS ? (((size_t)(nRows + 1) * 4 + 15) & ~(size_t)15) : 0));   // [nTrans]
  if (OFFLDS) for (int r = threadIdx.x; r <= nRows; r += blockDim.x) lOff[r] = m.inOff[r];
  for (int a = threadIdx.x; a < nTrans; a += blockDim.x) {
    const uint32_t eid = m.inEid[a];
    TbEdge e; e.w = m.inW[a]; e.eid = eid; e.src = (uint16_t)m.inSrc[a]; e.hasIn = m.eInTok[eid] != 0; e.hasOut = m.eOutTok[eid] != 0;
    lEdge[a] = e;
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= nPairs) return;
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen, S = m.S;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const double *cells = pool + pd.cellBase;
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  int i = inLen, o = outLen, s = S - 1;
  if (!(cells[((long long)o * I + i) * S + s] > -INFINITY)) { if (lane == 0) pathLen[p] = -1; return; }
  long long n = 0;
  uint32_t held = 0;
  const bool prefetch = 9 * S <= 128;
  double pfA = 0.0, pfB = 0.0;
  unsigned sink = 0;
  // tokens one step ahead: the next position consumes in[i - 1] (held) or in[i - 2] (requested now), likewise the output
  int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;
  while (i > 0 || o > 0 || s != 0) {
    const int itP = i > 1 ? in[i - 2] : 0, otP = o > 1 ? out[o - 2] : 0;
    const double *cur = cells + ((long long)o * I + i) * S;
    double best = -INFINITY; int bestIdx = 0x7fffffff; int bestA = -1;
    int base = 0;
    for (int grp = 0; grp < 4; ++grp) {
      int kit, kot; const double *sc;
      if (grp == 0) { if (!(i && o)) continue; kit = it; kot = ot; sc = cur - (I + 1) * S; }
      else if (grp == 1) { if (!i) continue; kit = it; kot = 0; sc = cur - S; }
      else if (grp == 2) { if (!o) continue; kit = 0; kot = ot; sc = cur - I * S; }
      else { kit = 0; kot = 0; sc = cur; }
      const int row = (s * (m.nIn + 1) + kit) * (m.nOut + 1) + kot;
      const int a0 = OFFLDS ? lOff[row] : m.inOff[row], a1 = OFFLDS ? lOff[row + 1] : m.inOff[row + 1];
      for (int a = a0 + lane; a < a1; a += 64) {
        const double v = sc[lEdge[a].src] + lEdge[a].w;
        const int idx = base + (a - a0);
        if (bestA < 0 || v > best || (v == best && idx < bestIdx)) { best = v; bestIdx = idx; bestA = a; }
      }
      base += a1 - a0;
    }
    if (prefetch) {
      // touch the supercells (i-2..i, o-2..o): lane L reads double L of the 3 x 3 x S block (two passes), clamped
      sink ^= (unsigned)__double2loint(pfA) ^ (unsigned)__double2loint(pfB);      // consumes LAST step's touches: no wait on this step's
      const int per = 3 * S;
      for (int k = 0; k < 2; ++k) {
        const int e = lane + 64 * k;
        const int r = min(e / per, 2), c = e - (e / per) * per;
        const long long po = max(o - r, 0), pi0 = max(i - 2, 0);
        const long long off = (po * I + pi0) * S + min(c, (int)((i - pi0 + 1) * S) - 1);
        const double v = cells[off];
        if (k == 0) pfA = v; else pfB = v;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bestIdx, off), oa = __shfl_xor(bestA, off);
      const bool take = oa >= 0 && (bestA < 0 || ov > best || (ov == best && oi < bestIdx));
      if (take) { best = ov; bestIdx = oi; bestA = oa; }
    }
    if (bestA < 0) { if (lane == 0) pathLen[p] = -3; return; }
    if (n >= cap) { if (lane == 0) pathLen[p] = -2; return; }
    const TbEdge be = lEdge[bestA];
    if ((int)(n & 63) == lane) held = be.eid;   // one store per 64 steps, see k_traceback
    ++n;
    if ((n & 63) == 0) pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = held;
    if (be.hasIn) { --i; it = itP; }
    if (be.hasOut) { --o; ot = otP; }
    s = (int)be.src;
  }
  if ((n & ~63ll) + lane < n) pathBuf[slot0 + cap - 1 - ((n & ~63ll) + lane)] = held;
  if (lane == 0) pathLen[p] = (sink == 0x9e3779b9u && n < 0) ? -4 : n;   // `sink` keeps the touches alive; never true
}

// Machines with large alphabets (psw2dna: 271 states x 105 label keys = 28 455 CSR rows for 1 684 edges): the row offsets do
// not fit LDS, but a state's incoming edges are CONTIGUOUS in the `incoming` view, sorted by label key.  The lanes scan the
// state's whole edge range from LDS and keep the edges whose key is one of the four the position allows (match, input-only,
// output-only, silent); (group, position) orders the candidates exactly as the reference enumerates them.  One trip to
// memory per step (the cells), tokens one step ahead, path stored 64 edges at a time.
struct TbEdgeK { double w; uint32_t eid; uint16_t src; uint16_t key; };
__global__ __launch_bounds__(256) void k_traceback_scan(DevMachine m, const PairDesc *__restrict__ pairs, long long nPairs,
                                                        long long nTrans, const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                        const double *__restrict__ pool, const long long *__restrict__ slotOff,
                                                        uint32_t *__restrict__ pathBuf, long long *__restrict__ pathLen) {
  extern __shared__ unsigned char tbs_raw[];
  TbEdgeK *lEdge = (TbEdgeK *)tbs_raw;                                  // [nTrans]
  int *sBeg = (int *)(tbs_raw + (size_t)nTrans * sizeof(TbEdgeK));      // [S + 1]: first incoming edge of each state
  const int S = m.S, K = m.K, NO = m.nOut + 1;
  for (int st = threadIdx.x; st <= S; st += blockDim.x) sBeg[st] = m.inOff[(long long)st * K];
  for (int a = threadIdx.x; a < nTrans; a += blockDim.x) {
    const uint32_t eid = m.inEid[a];
    TbEdgeK e; e.w = m.inW[a]; e.eid = eid; e.src = (uint16_t)m.inSrc[a]; e.key = (uint16_t)((int)m.eInTok[eid] * NO + (int)m.eOutTok[eid]);
    lEdge[a] = e;
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= nPairs) return;
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const double *cells = pool + pd.cellBase;
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  int i = inLen, o = outLen, s = S - 1;
  if (!(cells[((long long)o * I + i) * S + s] > -INFINITY)) { if (lane == 0) pathLen[p] = -1; return; }
  long long n = 0;
  uint32_t held = 0;
  int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;
  while (i > 0 || o > 0 || s != 0) {
    const int itP = i > 1 ? in[i - 2] : 0, otP = o > 1 ? out[o - 2] : 0;
    const double *cur = cells + ((long long)o * I + i) * S;
    const int kM = (i && o) ? it * NO + ot : -1, kI = i ? it * NO : -1, kO = o ? ot : -1;   // keys of the groups the position allows (silent: 0)
    double best = -INFINITY; int bestIdx = 0x7fffffff; int bestA = -1;
    const int b0 = sBeg[s], b1 = sBeg[s + 1];
    for (int a = b0 + lane; a < b1; a += 64) {
      const TbEdgeK e = lEdge[a];
      const int key = (int)e.key;
      int grp; const double *sc;
      if (key == kM) { grp = 0; sc = cur - (I + 1) * S; }
      else if (key == kI) { grp = 1; sc = cur - S; }
      else if (key == kO) { grp = 2; sc = cur - I * S; }
      else if (key == 0) { grp = 3; sc = cur; }
      else continue;
      const double v = sc[e.src] + e.w;
      const int idx = (grp << 24) | (a - b0);
      if (bestA < 0 || v > best || (v == best && idx < bestIdx)) { best = v; bestIdx = idx; bestA = a; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bestIdx, off), oa = __shfl_xor(bestA, off);
      const bool take = oa >= 0 && (bestA < 0 || ov > best || (ov == best && oi < bestIdx));
      if (take) { best = ov; bestIdx = oi; bestA = oa; }
    }
    if (bestA < 0) { if (lane == 0) pathLen[p] = -3; return; }
    if (n >= cap) { if (lane == 0) pathLen[p] = -2; return; }
    const TbEdgeK be = lEdge[bestA];
    if ((int)(n & 63) == lane) held = be.eid;   // one store per 64 steps, see k_traceback
    ++n;
    if ((n & 63) == 0) pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = held;
    const int bk = (int)be.key;
    if (bk / NO) { --i; it = itP; }
    if (bk % NO) { --o; ot = otP; }
    s = (int)be.src;
  }
  if ((n & ~63ll) + lane < n) pathBuf[slot0 + cap - 1 - ((n & ~63ll) + lane)] = held;
  if (lane == 0) pathLen[p] = n;
}

// ---- launch helpers (host) -----------------------------------------------------------------------------------
__global__ void k_fill_neg_inf(double *p, long long n) {
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) p[k] = -INFINITY;
}

int launch_fill_neg_inf(double *d_pool, long long n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_fill_neg_inf, dim3((unsigned)std::min<long long>((n + 255) / 256, 65536)), dim3(256), 0, st, d_pool, n);
  return hip_ok(hipGetLastError(), "fill launch") ? 0 : 1;
}

int launch_generic_fill(const mb_machine *m, int mode, const PairDesc *d_pairs, long long nPairs, const int *d_in,
                        const int *d_out, double *d_pool, int startState, hipStream_t st, const int *d_envStart,
                        const int *d_envEnd) {
  if (nPairs == 0) return 0;
  ++g_last_launches;
  const dim3 grid((unsigned)nPairs), block(m->S * 64 >= 1024 ? 1024 : 256);
  if (mode == MB_FORWARD)
    hipLaunchKernelGGL(k_generic_fill_fwd<MB_FORWARD>, grid, block, 0, st, m->dev, d_pairs, d_in, d_out, d_pool, startState, d_envStart, d_envEnd);
  else if (mode == MB_VITERBI)
    hipLaunchKernelGGL(k_generic_fill_fwd<MB_VITERBI>, grid, block, 0, st, m->dev, d_pairs, d_in, d_out, d_pool, 0, d_envStart, d_envEnd);
  else
    hipLaunchKernelGGL(k_generic_fill_bwd, grid, block, 0, st, m->dev, d_pairs, d_in, d_out, d_pool, d_envStart, d_envEnd);
  return hip_ok(hipGetLastError(), "generic fill launch") ? 0 : 1;
}

int launch_gather_loglike(const PairDesc *d_pairs, long long nPairs, const double *d_pool, int S, int backward,
                          double *d_loglike, hipStream_t st) {
  if (nPairs == 0) return 0;
  hipLaunchKernelGGL(k_gather_loglike, dim3((unsigned)((nPairs + 255) / 256)), dim3(256), 0, st, d_pairs, nPairs, d_pool, S,
                     backward, d_loglike);
  return hip_ok(hipGetLastError(), "gather launch") ? 0 : 1;
}

int launch_generic_counts(const mb_machine *m, const PairDesc *d_pairs, long long nPairs, long long maxPairCells,
                          const int *d_in, const int *d_out, const double *d_fwd, const double *d_bwd, double *d_counts,
                          hipStream_t st) {
  if (nPairs == 0) return 0;
  // enough workgroups to fill the chip (256 CUs x 8), bounded by the work one pair offers
  long long chunks = (2048 + nPairs - 1) / nPairs;
  const long long maxChunks = (maxPairCells + 255) / 256;
  if (chunks > maxChunks) chunks = maxChunks;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(k_generic_counts, dim3((unsigned)(nPairs * chunks)), dim3(256), 0, st, m->dev, d_pairs, d_in, d_out,
                     d_fwd, d_bwd, m->nTrans, (int)chunks, d_counts, g_deterministic ? 1 : 0);
  return hip_ok(hipGetLastError(), "counts launch") ? 0 : 1;
}

// Paths are written backwards from the END of each pair's slot (its worst-case length); this packs them back to back in
// start -> end order so that one D2H of exactly the used bytes lands in the caller's array.  off[p] < 0: no path.
__global__ __launch_bounds__(256) void k_compact_paths(const uint32_t *__restrict__ pathBuf, const long long *__restrict__ slotOff,
                                                       const long long *__restrict__ pathLen, const long long *__restrict__ off,
                                                       uint32_t *__restrict__ out) {
  const long long p = blockIdx.x, n = pathLen[p];
  if (n <= 0) return;
  const uint32_t *src = pathBuf + slotOff[p + 1] - n;
  uint32_t *dst = out + off[p];
  for (long long k = threadIdx.x; k < n; k += blockDim.x) dst[k] = src[k];
}

int launch_compact_paths(const uint32_t *d_pathBuf, const long long *d_slotOff, const long long *d_pathLen, const long long *d_off,
                         uint32_t *d_out, long long nPairs, hipStream_t st) {
  if (nPairs == 0) return 0;
  hipLaunchKernelGGL(k_compact_paths, dim3((unsigned)nPairs), dim3(256), 0, st, d_pathBuf, d_slotOff, d_pathLen, d_off, d_out);
  return hip_ok(hipGetLastError(), "path compaction launch") ? 0 : 1;
}

// DPMatrix::traceBack with selectMaxTrans over ONE TRACEBACK BYTE per cell (tiled family, MED_MODE_TB): byte (i, o, s) =
// table << 6 | index, the first maximal candidate of the cell in the reference's enumeration order (tables: 0 match, 1 input-
// only, 2 output-only, 3 silent; index: position in that label's `incoming` list, src/dpmatrix.defs.h:93-103) -- the choice
// std::max_element makes there.  No candidate is re-evaluated.  One wavefront per pair, four pairs per workgroup sharing the
// edges (state-major, sorted by label key, so a label's list is contiguous) in LDS.  The bytes of the supercell the walk
// stands on sit in LDS (silent moves cost no memory access); on arrival the three supercells it can move to are requested
// at once, so an emitting move waits for at most what is left of one trip to memory after the silent chain in between.
struct TbEdgeB { uint32_t eid; uint16_t src; uint16_t key; };
template <int R>
__global__ __launch_bounds__(256) void k_traceback_bytes(DevMachine m, const PairDesc *__restrict__ pairs, long long nPairs, long long nTrans,
                                                         const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                         const unsigned char *__restrict__ tb, int Sb, const double *__restrict__ ll,
                                                         const long long *__restrict__ slotOff, uint32_t *__restrict__ pathBuf,
                                                         long long *__restrict__ pathLen) {
  extern __shared__ unsigned char tbb_raw[];
  const int S = m.S, K = m.K, NO = m.nOut + 1;
  TbEdgeB *lEdge = (TbEdgeB *)tbb_raw;                                                  // [nTrans]
  int *sBeg = (int *)(tbb_raw + (size_t)nTrans * sizeof(TbEdgeB));                      // [S + 1]: first incoming edge of each state
  unsigned char *cellAll = tbb_raw + ((((size_t)nTrans * sizeof(TbEdgeB) + (size_t)(S + 1) * 4) + 15) & ~(size_t)15);
  for (int st = threadIdx.x; st <= S; st += blockDim.x) sBeg[st] = m.inOff[(long long)st * K];
  for (int a = threadIdx.x; a < nTrans; a += blockDim.x) {
    const uint32_t eid = m.inEid[a];
    TbEdgeB e; e.eid = eid; e.src = (uint16_t)m.inSrc[a]; e.key = (uint16_t)((int)m.eInTok[eid] * NO + (int)m.eOutTok[eid]);
    lEdge[a] = e;
  }
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + wv;
  if (p >= nPairs) return;
  if (!(ll[p] > -INFINITY)) { if (lane == 0) pathLen[p] = -1; return; }
  unsigned char *cellB = cellAll + (size_t)wv * (size_t)(R * 1024);                    // this walk's current supercell
  const PairDesc pd = pairs[p];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const long long I = inLen + 1;
  const int *in = inTok + pd.inBase, *out = outTok + pd.outBase;
  const unsigned char *bytes = tb + pd.cellBase;
  const long long slot0 = slotOff[p], cap = slotOff[p + 1] - slot0;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  struct SC { u4 r[R]; };
  auto loadSC = [&](int ci, int co) -> SC {     // clamped, unconditional: cells outside the lattice are never moved into
    SC c;
    const unsigned char *q = bytes + ((long long)max(co, 0) * I + max(ci, 0)) * Sb;
#pragma unroll
    for (int k = 0; k < R; ++k) c.r[k] = *(const u4 *)(q + min((k * 64 + lane) * 16, Sb - 16));
    return c;
  };
  auto stash = [&](const SC &c) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) if ((k * 64 + lane) * 16 < Sb) *(u4 *)(cellB + (k * 64 + lane) * 16) = c.r[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  int i = inLen, o = outLen, s = S - 1;
  stash(loadSC(i, o));
  SC nd = loadSC(i - 1, o - 1), nl = loadSC(i - 1, o), nu = loadSC(i, o - 1);
  int it = i ? in[i - 1] : 0, ot = o ? out[o - 1] : 0;
  int itP = i > 1 ? in[i - 2] : 0, otP = o > 1 ? out[o - 2] : 0;
  long long n = 0;
  uint32_t held = 0;
  while (i > 0 || o > 0 || s != 0) {
    const unsigned code = cellB[s];
    const int T = (int)(code >> 6), j = (int)(code & 63u);
    const int want = T == 0 ? it * NO + ot : (T == 1 ? it * NO : (T == 2 ? ot : 0));
    const int b0 = sBeg[s], b1 = sBeg[s + 1];
    int a = -1;
    for (int base = b0; base < b1; base += 64) {       // the label's list is contiguous: its first edge + j
      const int idx = base + lane;
      const unsigned long long mask = __ballot(idx < b1 && (int)lEdge[min(idx, b1 - 1)].key == want);
      if (mask) { a = base + __builtin_ctzll(mask) + j; break; }
    }
    if (a < 0 || a >= b1) { if (lane == 0) pathLen[p] = -3; return; }
    if (n >= cap) { if (lane == 0) pathLen[p] = -2; return; }
    const TbEdgeB be = lEdge[a];
    if ((int)(n & 63) == lane) held = be.eid;          // one store per 64 steps (a store counts in vmcnt like a load on gfx9)
    ++n;
    if ((n & 63) == 0) pathBuf[slot0 + cap - 1 - (n - 64 + lane)] = held;
    s = (int)be.src;
    if (T != 3) {
      if (T == 0) { stash(nd); --i; --o; it = itP; ot = otP; }
      else if (T == 1) { stash(nl); --i; it = itP; }
      else { stash(nu); --o; ot = otP; }
      nd = loadSC(i - 1, o - 1); nl = loadSC(i - 1, o); nu = loadSC(i, o - 1);
      itP = i > 1 ? in[i - 2] : 0; otP = o > 1 ? out[o - 2] : 0;
    }
  }
  if ((n & ~63ll) + lane < n) pathBuf[slot0 + cap - 1 - ((n & ~63ll) + lane)] = held;
  if (lane == 0) pathLen[p] = n;
}

// bytes of LDS the kernel needs (edges + state offsets + four supercells); 0 = the machine does not fit
size_t traceback_bytes_lds(const mb_machine *m, int Sb) {
  if (Sb > 4096 || m->S > 65535 || m->K > 65535) return 0;
  const int R = Sb <= 1024 ? 1 : (Sb <= 2048 ? 2 : 4);
  const size_t b = ((((size_t)m->nTrans * sizeof(TbEdgeB) + (size_t)(m->S + 1) * 4) + 15) & ~(size_t)15) + 4 * (size_t)R * 1024;
  return b <= 150 * 1024 ? b : 0;
}

int launch_traceback_bytes(const mb_machine *m, const PairDesc *d_pairs, long long nPairs, const int *d_in, const int *d_out,
                           const unsigned char *d_tb, int Sb, const double *d_ll, const long long *d_slotOff, uint32_t *d_pathBuf,
                           long long *d_pathLen, hipStream_t st) {
  if (nPairs == 0) return 0;
  const size_t lds = traceback_bytes_lds(m, Sb);
  if (!lds) { set_error("traceback bytes: machine does not fit the LDS tables"); return 1; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)&k_traceback_bytes<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)&k_traceback_bytes<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)&k_traceback_bytes<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)((nPairs + 3) / 4)), block(256);
  if (Sb <= 1024) hipLaunchKernelGGL(k_traceback_bytes<1>, grid, block, lds, st, m->dev, d_pairs, nPairs, (long long)m->nTrans, d_in, d_out, d_tb, Sb, d_ll, d_slotOff, d_pathBuf, d_pathLen);
  else if (Sb <= 2048) hipLaunchKernelGGL(k_traceback_bytes<2>, grid, block, lds, st, m->dev, d_pairs, nPairs, (long long)m->nTrans, d_in, d_out, d_tb, Sb, d_ll, d_slotOff, d_pathBuf, d_pathLen);
  else hipLaunchKernelGGL(k_traceback_bytes<4>, grid, block, lds, st, m->dev, d_pairs, nPairs, (long long)m->nTrans, d_in, d_out, d_tb, Sb, d_ll, d_slotOff, d_pathBuf, d_pathLen);
  return hip_ok(hipGetLastError(), "traceback (bytes) launch") ? 0 : 1;
}

int launch_traceback(const mb_machine *m, const PairDesc *d_pairs, long long nPairs, const int *d_in, const int *d_out,
                     const double *d_pool, const long long *d_slotOff, uint32_t *d_pathBuf, long long *d_pathLen,
                     hipStream_t st) {
  if (nPairs == 0) return 0;
  const long long nRows = (long long)m->S * m->K;
  const size_t ldsBytes = (((size_t)(nRows + 1) * 4 + 15) & ~(size_t)15) + (size_t)m->nTrans * sizeof(TbEdge);
  static int useLds = -1;
  if (useLds < 0) { const char *e = opt_env("MB_TRACEBACK_LDS"); useLds = (e && *e == '0') ? 0 : 1; }
  if (useLds && nRows <= 8192 && m->nTrans <= 2048 && m->S <= 65535 && ldsBytes <= 64 * 1024) {
    hipLaunchKernelGGL(k_traceback_lds<true>, dim3((unsigned)((nPairs + 3) / 4)), dim3(256), ldsBytes, st, m->dev, d_pairs, nPairs,
                       (long long)m->nTrans, d_in, d_out, d_pool, d_slotOff, d_pathBuf, d_pathLen);
    return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
  }
  const size_t scanBytes = (size_t)m->nTrans * sizeof(TbEdgeK) + (size_t)(m->S + 1) * 4;   // edges (<= 56 KB) + one offset per state
  if (useLds && m->nTrans <= 3584 && m->S <= 8192 && m->K <= 65535 && env_int_g("MB_TRACEBACK_SCAN", 1)) {
    static bool attr = false;   // up to 56 KB + 32 KB: beyond the 64 KB a kernel may use without asking
    if (!attr) { (void)hipFuncSetAttribute((const void *)&k_traceback_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(k_traceback_scan, dim3((unsigned)((nPairs + 3) / 4)), dim3(256), scanBytes, st,
                       m->dev, d_pairs, nPairs, (long long)m->nTrans, d_in, d_out, d_pool, d_slotOff, d_pathBuf, d_pathLen);
    return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
  }
  if (useLds && m->nTrans <= 3584 && m->S <= 65535) {   // edges only (<= 56 KB): the offsets are read from global memory
    hipLaunchKernelGGL(k_traceback_lds<false>, dim3((unsigned)((nPairs + 3) / 4)), dim3(256), (size_t)m->nTrans * sizeof(TbEdge), st, m->dev,
                       d_pairs, nPairs, (long long)m->nTrans, d_in, d_out, d_pool, d_slotOff, d_pathBuf, d_pathLen);
    return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
  }
  hipLaunchKernelGGL(k_traceback, dim3((unsigned)nPairs), dim3(64), 0, st, m->dev, d_pairs, d_in, d_out, d_pool, d_slotOff,
                     d_pathBuf, d_pathLen);
  return hip_ok(hipGetLastError(), "traceback launch") ? 0 : 1;
}

}  // namespace mb

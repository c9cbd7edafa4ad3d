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
#include <cstring>
#include <numeric>

#include "mb_internal.h"
#include "mb_device_math.h"
#include "mb_medium.h"

namespace mb {

// ------------------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void acc_sum(double &m, float &s, double v) {
  const double d64 = (v == m) ? 0.0 : (v - m);       // (-inf) - (-inf) guarded like src/logsumexp.h:79
  const float d = (float)d64;
  const float e = __builtin_amdgcn_exp2f(-fabsf(d) * 1.44269504088896f);
  const bool up = d > 0.0f;
  s = up ? fmaf(s, e, 1.0f) : (s + e);
  m = up ? v : m;
}

__device__ __forceinline__ double fin_sum(double m, float s) {
  return m + (double)(__builtin_amdgcn_logf(s) * 0.693147180559945f);
}

template <int MODE>
struct Acc {
  double m; float s;
  __device__ __forceinline__ void init() { m = -INFINITY; s = 0.0f; }
  __device__ __forceinline__ void add(double v) {
    if (MODE == MB_VITERBI) m = dmax(m, v); else acc_sum(m, s, v);
  }
  __device__ __forceinline__ double result() const { return MODE == MB_VITERBI ? m : fin_sum(m, s); }
};

struct MedTileArgs {
  const PairDesc *pairs;
  const int *inTok, *outTok;
  double *pool;               // materialised matrices, reference layout; nullptr in rolling mode
  double *colHalo;            // rolling mode: per pair two [outLen+1][S] column buffers (ping-pong by strip parity)
  const long long *haloBase;  // rolling mode: per pair offset (in doubles) of its two buffers
  double *loglike;            // rolling mode: loglike[pair]
  int C, TS, launch, rev, materialise, startNode;
};

template <int MODE, int G>
__global__ __launch_bounds__(1024) void k_medium_tile(MedProgDev P, MedTileArgs A) {
  extern __shared__ double lds[];
  constexpr int LPG = 64 / G;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane / LPG, q = lane - g * LPG;
  const int S = P.S, Spad = P.Spad, NS = P.NS, C = A.C;
  const PairDesc pd = A.pairs[blockIdx.y];
  const int inLen = pd.inLen, outLen = pd.outLen;
  const long long I = inLen + 1;
  // which tile
  int a, b;
  if (A.materialise) { a = blockIdx.x; b = A.launch - 2 * a; } else { a = A.launch; b = 0; }
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
    double *hb = A.colHalo + A.haloBase[blockIdx.y];
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

  // ---- preload the ring state of steps t0-1 (and t0-2 when match edges exist) --------------------------------
  for (int dt = 1; dt < NS; ++dt) {
    const int tp = t0 - dt;
    const int slot = ((tp % NS) + NS) % NS;
    for (int col = 0; col <= C; ++col) {
      const int cc = col - 1, ci = i0 + cc, co = tp - cc;
      if (ci < 0 || ci > inLen || co < 0 || co > outLen) continue;
      const double *src = nullptr;
      if (A.materialise) src = cellPtr(ci, co);
      else if (cc == -1) src = haloIn + (long long)co * S;
      if (!src) continue;
      double *dstp = ring(slot, col);
      for (int j = tid; j < S; j += blockDim.x) dstp[j] = src[j];
    }
  }
  __syncthreads();

  int slotCur = t0 % NS;
  for (int t = t0; t < t1; ++t) {
    const int o = t - c;
    const bool active = colValid && o >= 0 && o <= outLen;
    const int ot = (active && o > 0) ? (rev ? out[outLen - o] : out[o - 1]) : 0;
    const int slotPrev = (slotCur + NS - 1) % NS, slotPrev2 = (slotCur + NS - 2) % NS;
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
    double *cur = ring(slotCur, c + 1);
    const double *down = ring(slotPrev, c + 1), *left = ring(slotPrev, c), *diag = ring(slotPrev2, c);
    const bool origin = (i == 0 && o == 0);
    for (int r = 0; r < P.R; ++r) {
      const int d = P.dest[r * LPG + q];
      Acc<MODE> acc;
      acc.init();
      if (origin && d == A.startNode) acc.add(0.0);
      {  // match: (i-1,o-1)
        const int ns = P.tab[0].nslots[r];
        if (ns) {
          const bool ok = active && i > 0 && o > 0;
          const int base = P.tab[0].base[r] + ((it * (P.nOut + 1) + ot) * ns) * LPG + q;
          for (int k = 0; k < ns; ++k) {
            const int sidx = P.tab[0].src[base + k * LPG];
            const double w = P.tab[0].w[base + k * LPG];
            const double v = ok ? diag[sidx] + w : -INFINITY;
            acc.add(v);
          }
        }
      }
      {  // input-only: (i-1,o)
        const int ns = P.tab[1].nslots[r];
        if (ns) {
          const bool ok = active && i > 0;
          const int base = P.tab[1].base[r] + (it * ns) * LPG + q;
          for (int k = 0; k < ns; ++k) {
            const int sidx = P.tab[1].src[base + k * LPG];
            const double w = P.tab[1].w[base + k * LPG];
            const double v = ok ? left[sidx] + w : -INFINITY;
            acc.add(v);
          }
        }
      }
      {  // output-only: (i,o-1)
        const int ns = P.tab[2].nslots[r];
        if (ns) {
          const bool ok = active && o > 0;
          const int base = P.tab[2].base[r] + (ot * ns) * LPG + q;
          for (int k = 0; k < ns; ++k) {
            const int sidx = P.tab[2].src[base + k * LPG];
            const double w = P.tab[2].w[base + k * LPG];
            const double v = ok ? down[sidx] + w : -INFINITY;
            acc.add(v);
          }
        }
      }
      {  // silent: same supercell, lower levels
        const int ns = P.tab[3].nslots[r];
        if (ns) {
          const int base = P.tab[3].base[r] + q;
          for (int k = 0; k < ns; ++k) {
            const int sidx = P.tab[3].src[base + k * LPG];
            const double w = P.tab[3].w[base + k * LPG];
            const double v = active ? cur[sidx] + w : -INFINITY;
            acc.add(v);
          }
        }
      }
      if (active && d >= 0) cur[d] = acc.result();
      if (P.sync[r]) {  // wave-local: the next round reads what other lanes of this wave just wrote
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- copy the finished supercell out ----------------------------------------------------------------------
    if (active) {
      if (A.materialise) {
        double *dstp = cellPtr(i, o);
        for (int j = q; j < S; j += LPG) dstp[j] = cur[j];
      } else {
        if (c == C - 1) {
          double *dstp = haloOut + (long long)o * S;
          for (int j = q; j < S; j += LPG) dstp[j] = cur[j];
        }
        if (i == inLen && o == outLen && q == 0) A.loglike[blockIdx.y] = cur[P.endNode];
      }
    }
    if (wantHalo) {
      double *hd = ring(slotCur, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = tid + k * blockDim.x;
        if (j < S) hd[j] = hv[k];
      }
    }
    __syncthreads();
    slotCur = (slotCur + 1) % NS;
  }
}

// ------------------------------------------------------------------------------------------------------------
// host side: program compiler
// ------------------------------------------------------------------------------------------------------------
struct Cand { uint16_t src; uint32_t eid; };

static void build_program(const mb_machine *m, bool backward, int G, MedProgram &P) {
  const int S = m->S, LPG = 64 / G, nIn = m->nIn, nOut = m->nOut;
  P.G = G; P.LPG = LPG; P.backward = backward;
  P.Spad = (S + 1) & ~1;
  P.NS = m->hasMatch ? 3 : 2;
  const std::vector<int> &lev = backward ? m->levB : m->levF;
  const std::vector<int> &off = backward ? m->outOff : m->inOff;
  const std::vector<uint32_t> &perm = backward ? m->outPerm : m->inPerm;
  const int nLev = backward ? m->nLevB : m->nLevF;
  auto other = [&](uint32_t e) { return backward ? m->dst[e] : m->src[e]; };
  auto row = [&](int st, int it, int ot) { return ((long long)st * (nIn + 1) + it) * (nOut + 1) + ot; };
  // candidate list of (state, table, token) in the reference's iteration order
  auto cands = [&](int st, int T, int tok, std::vector<Cand> &out) {
    out.clear();
    int it = 0, ot = 0;
    if (T == 0) { it = tok / (nOut + 1); ot = tok % (nOut + 1); if (!it || !ot) return; }
    else if (T == 1) { it = tok; if (!it) return; }
    else if (T == 2) { ot = tok; if (!ot) return; }
    const long long rw = row(st, it, ot);
    for (int a = off[rw]; a < off[rw + 1]; ++a) {
      const uint32_t e = perm[a];
      const uint32_t o = other(e);
      if (T == 3 && (backward ? o <= (uint32_t)st : o >= (uint32_t)st)) continue;  // silent self-loop on state 0
      out.push_back({(uint16_t)o, e});
    }
  };
  const int ntok[4] = {(nIn + 1) * (nOut + 1), nIn + 1, nOut + 1, 1};
  // rounds: per level, states sorted by silent degree (descending) so that a round's slot count is tight
  std::vector<Cand> tmp;
  std::vector<std::vector<int>> rounds;
  std::vector<unsigned char> sync;
  for (int l = 0; l < nLev; ++l) {
    std::vector<int> st;
    for (int s = 0; s < S; ++s) if (lev[s] == l) st.push_back(s);
    std::vector<int> deg(S, 0);
    for (int s : st) {
      cands(s, 3, 0, tmp); deg[s] = (int)tmp.size() * 1000;
      int mx = 0;
      for (int T = 0; T < 3; ++T) for (int tok = 0; tok < ntok[T]; ++tok) { cands(s, T, tok, tmp); mx = std::max(mx, (int)tmp.size()); }
      deg[s] += mx;
    }
    std::stable_sort(st.begin(), st.end(), [&](int x, int y) { return deg[x] > deg[y]; });
    for (size_t k = 0; k < st.size(); k += LPG) {
      rounds.emplace_back(st.begin() + k, st.begin() + std::min(st.size(), k + LPG));
      sync.push_back(0);
    }
    if (!sync.empty()) sync.back() = 1;
  }
  if (!sync.empty()) sync.back() = 0;
  const int R = (int)rounds.size();
  P.R = R; P.sync = sync;
  P.dest.assign((size_t)R * LPG, -1);
  for (int r = 0; r < R; ++r) for (size_t k = 0; k < rounds[r].size(); ++k) P.dest[(size_t)r * LPG + k] = (short)rounds[r][k];
  for (int T = 0; T < 4; ++T) {
    P.nslots[T].assign(R, 0); P.base[T].assign(R, 0);
    P.src[T].clear(); P.eid[T].clear();
    for (int r = 0; r < R; ++r) {
      int ns = 0;
      for (int s : rounds[r]) for (int tok = 0; tok < ntok[T]; ++tok) { cands(s, T, tok, tmp); ns = std::max(ns, (int)tmp.size()); }
      P.nslots[T][r] = ns; P.base[T][r] = (int)P.src[T].size();
      if (!ns) continue;
      const size_t sz = (size_t)ntok[T] * ns * LPG, b0 = P.src[T].size();
      P.src[T].resize(b0 + sz, 0); P.eid[T].resize(b0 + sz, 0xFFFFFFFFu);
      for (size_t k = 0; k < rounds[r].size(); ++k)
        for (int tok = 0; tok < ntok[T]; ++tok) {
          cands(rounds[r][k], T, tok, tmp);
          for (size_t j = 0; j < tmp.size(); ++j) {
            const size_t idx = b0 + ((size_t)tok * ns + j) * LPG + k;
            P.src[T][idx] = tmp[j].src; P.eid[T][idx] = tmp[j].eid;
          }
        }
    }
  }
}

template <class T>
static bool up(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D(program)")) return false;
  return true;
}

bool medium_refresh_weights(const mb_machine *m, MedProgram &P) {
  for (int T = 0; T < 4; ++T) {
    std::vector<double> w(P.eid[T].size());
    for (size_t k = 0; k < w.size(); ++k) w[k] = P.eid[T][k] == 0xFFFFFFFFu ? -INFINITY : m->logW[P.eid[T][k]];
    if (!up(P.d_w[T], w)) return false;
    P.dev.tab[T].w = P.d_w[T];
  }
  return true;
}

bool medium_build(const mb_machine *m, bool backward, int G, MedProgram &P) {
  build_program(m, backward, G, P);
  if (!up(P.d_dest, P.dest) || !up(P.d_sync, P.sync)) return false;
  for (int T = 0; T < 4; ++T)
    if (!up(P.d_src[T], P.src[T]) || !up(P.d_base[T], P.base[T]) || !up(P.d_nslots[T], P.nslots[T])) return false;
  MedProgDev &d = P.dev;
  d.S = m->S; d.Spad = P.Spad; d.R = P.R; d.LPG = P.LPG; d.G = G; d.NS = P.NS;
  d.nIn = m->nIn; d.nOut = m->nOut;
  d.startNode = backward ? m->S - 1 : 0; d.endNode = backward ? 0 : m->S - 1;
  d.dest = P.d_dest; d.sync = P.d_sync;
  for (int T = 0; T < 4; ++T) { d.tab[T].src = P.d_src[T]; d.tab[T].base = P.d_base[T]; d.tab[T].nslots = P.d_nslots[T]; }
  return medium_refresh_weights(m, P);
}

void medium_free(MedProgram &P) {
  void *ptrs[] = {P.d_dest, P.d_sync};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (int T = 0; T < 4; ++T) {
    void *q[] = {P.d_src[T], P.d_w[T], P.d_base[T], P.d_nslots[T]};
    for (void *p : q) if (p) (void)hipFree(p);
  }
  P = MedProgram();
}

// Geometry: columns per strip limited by the 160 KB LDS of a CU.
bool medium_geometry(const mb_machine *m, const MedProgram &P, MedGeom &geo) {
  const size_t perCol = (size_t)P.NS * P.Spad * sizeof(double);
  const size_t budget = 160 * 1024 - 1024;
  long long maxCols = (long long)(budget / perCol) - 1;   // one extra column for the halo
  if (maxCols < P.G) return false;
  int waves = (int)std::min<long long>(maxCols / P.G, 16);
  // S must be covered by 4 halo registers per thread
  while (waves < 16 && (long long)waves * 64 * 4 < m->S) ++waves;
  if ((long long)waves * 64 * 4 < m->S || (long long)waves * P.G > maxCols) return false;
  geo.waves = waves; geo.C = waves * P.G;
  geo.ldsBytes = (size_t)P.NS * (geo.C + 1) * P.Spad * sizeof(double);
  return true;
}

template <int MODE>
static void launch_tile(int G, dim3 grid, dim3 block, size_t ldsBytes, hipStream_t st, const MedProgDev &P, const MedTileArgs &A) {
  switch (G) {
    case 1: hipLaunchKernelGGL((k_medium_tile<MODE, 1>), grid, block, ldsBytes, st, P, A); break;
    case 2: hipLaunchKernelGGL((k_medium_tile<MODE, 2>), grid, block, ldsBytes, st, P, A); break;
    case 4: hipLaunchKernelGGL((k_medium_tile<MODE, 4>), grid, block, ldsBytes, st, P, A); break;
    default: hipLaunchKernelGGL((k_medium_tile<MODE, 8>), grid, block, ldsBytes, st, P, A); break;
  }
}

static bool g_attr_set = false;
static void set_lds_attr() {
  if (g_attr_set) return;
#define SET(M, GG) (void)hipFuncSetAttribute((const void *)k_medium_tile<M, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  SET(0, 1); SET(0, 2); SET(0, 4); SET(0, 8); SET(1, 1); SET(1, 2); SET(1, 4); SET(1, 8);
#undef SET
  g_attr_set = true;
}

// Materialised fill of a chunk of pairs: wavefront of parallelogram tiles, launch index = 2*strip + block.
int medium_fill_materialised(const mb_machine *m, const MedProgram &P, const MedGeom &geo, int mode, const PairDesc *d_pairs,
                             const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_pool,
                             hipStream_t st) {
  if (pairs.empty()) return 0;
  set_lds_attr();
  int maxIn = 0, maxOut = 0;
  for (const PairDesc &pd : pairs) { maxIn = std::max(maxIn, pd.inLen); maxOut = std::max(maxOut, pd.outLen); }
  const int C = geo.C;
  const int NA = (maxIn + C) / C;
  const int T = maxOut + C;
  // tile length: enough blocks that ~2 workgroups per CU are in flight on the widest wavefront, but >= C steps
  // (dependency (a-1,b+1) needs TS >= C) and >= 64 steps to amortise the preload.
  long long wantBlocks = std::max<long long>(1, (512 + (long long)pairs.size() - 1) / (long long)pairs.size());
  int TS = (int)std::max<long long>(std::max(C, 64), (T + 2 * wantBlocks - 1) / (2 * wantBlocks));
  if (NA == 1) TS = T;  // a single strip has no wavefront to exploit
  const int NB = (T + TS - 1) / TS;
  MedTileArgs A{};
  A.pairs = d_pairs; A.inTok = d_in; A.outTok = d_out; A.pool = d_pool; A.colHalo = nullptr; A.haloBase = nullptr;
  A.loglike = nullptr; A.C = C; A.TS = TS; A.rev = P.backward ? 1 : 0; A.materialise = 1; A.startNode = P.dev.startNode;
  const dim3 grid(NA, (unsigned)pairs.size()), block(geo.waves * 64);
  for (int launch = 0; launch <= 2 * (NA - 1) + (NB - 1); ++launch) {
    A.launch = launch;
    if (mode == MB_VITERBI) launch_tile<MB_VITERBI>(P.G, grid, block, geo.ldsBytes, st, P.dev, A);
    else launch_tile<MB_FORWARD>(P.G, grid, block, geo.ldsBytes, st, P.dev, A);
  }
  return hip_ok(hipGetLastError(), "medium tile launch") ? 0 : 1;
}

// Rolling (log-likelihood only) Forward: one workgroup per pair per launch, strips in sequence.
int medium_forward_rolling(const mb_machine *m, const MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs,
                           const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_colHalo,
                           const long long *d_haloBase, double *d_loglike, hipStream_t st) {
  if (pairs.empty()) return 0;
  set_lds_attr();
  int maxIn = 0, maxOut = 0;
  for (const PairDesc &pd : pairs) { maxIn = std::max(maxIn, pd.inLen); maxOut = std::max(maxOut, pd.outLen); }
  const int C = geo.C, NA = (maxIn + C) / C;
  MedTileArgs A{};
  A.pairs = d_pairs; A.inTok = d_in; A.outTok = d_out; A.pool = nullptr; A.colHalo = d_colHalo; A.haloBase = d_haloBase;
  A.loglike = d_loglike; A.C = C; A.TS = maxOut + C + 1; A.rev = 0; A.materialise = 0; A.startNode = P.dev.startNode;
  const dim3 grid(1, (unsigned)pairs.size()), block(geo.waves * 64);
  for (int a = 0; a < NA; ++a) {
    A.launch = a;
    launch_tile<MB_FORWARD>(P.G, grid, block, geo.ldsBytes, st, P.dev, A);
  }
  return hip_ok(hipGetLastError(), "medium rolling launch") ? 0 : 1;
}

}  // namespace mb

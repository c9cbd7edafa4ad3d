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
  double *loglike;            // rolling mode: loglike[pair]
  int C, TS, launch, rev, materialise, startNode;
};

#define MED_L2E 1.44269504088896f
#define MED_LN2 0.693147180559945f

template <int MODE, int G>
__global__ __launch_bounds__(1024) void k_medium_tile(MedProgDev P, MedTileArgs A) {
  extern __shared__ double lds[];
  constexpr int LPG = 64 / G;
  constexpr int MS = MED_MAXSLOT;
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
  int *ldsMeta = (int *)(lds + (long long)NS * (C + 1) * Spad);
  short *ldsDest = (short *)(ldsMeta + P.nChunks * (1 + MS));

  // ---- program metadata into LDS --------------------------------------------------------------------------------
  for (int j = tid; j < P.nChunks * (1 + MS); j += blockDim.x) ldsMeta[j] = P.meta[j];
  for (int j = tid; j < P.R * LPG; j += blockDim.x) ldsDest[j] = P.dest[j];
  // sentinel: padding candidates (source index S, weight -inf) must read -inf, never stale LDS (NaN + -inf = NaN)
  for (int j = tid; j < NS * (C + 1); j += blockDim.x) lds[(long long)j * Spad + S] = -INFINITY;
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

  const uint16_t *__restrict__ gsrc = P.src;
  const double *__restrict__ gw = P.w;
  int slotCur = t0 % NS;
  // output token of the first step
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
    const bool origin = active && (i == 0 && o == 0);
    const bool anyOrigin = __any(origin);
    // per-table token offsets and validity
    const int tokM = (it * (P.nOut + 1) + ot) * LPG + q, tokI = it * LPG + q, tokO = ot * LPG + q;
    const bool okM = active && i > 0 && o > 0, okI = active && i > 0, okO = active && o > 0;

    // software pipeline over chunks: candidates of chunk c+1 are fetched while chunk c is evaluated
    int cS[MS]; double cW[MS];
    int nS[MS]; double nW[MS];
    auto fetch = [&](int ch, int (&fs)[MS], double (&fw)[MS]) {
      const int m0 = __builtin_amdgcn_readfirstlane(ldsMeta[ch * (1 + MS)]);
      const int ns = m0 & 15;
#pragma unroll
      for (int k = 0; k < MS; ++k) {
        if (k < ns) {
          const unsigned sd = (unsigned)__builtin_amdgcn_readfirstlane(ldsMeta[ch * (1 + MS) + 1 + k]);
          const unsigned tb = sd >> 28;
          const int off = (int)(sd & 0x0FFFFFFFu);
          const int idx = off + (tb == 0 ? tokM : (tb == 1 ? tokI : (tb == 2 ? tokO : q)));
          fs[k] = gsrc[idx]; fw[k] = gw[idx];
        } else { fs[k] = S; fw[k] = -INFINITY; }
      }
    };
    fetch(0, cS, cW);
    double accM = -INFINITY; float accS = 0.0f;
    for (int ch = 0; ch < P.nChunks; ++ch) {
      const int m0 = __builtin_amdgcn_readfirstlane(ldsMeta[ch * (1 + MS)]);
      const int ns = m0 & 15, round = m0 >> 8;
      const bool first = (m0 >> 4) & 1, last = (m0 >> 5) & 1, sync = (m0 >> 6) & 1;
      if (ch + 1 < P.nChunks) fetch(ch + 1, nS, nW);
      const int d = ldsDest[round * LPG + q];
      // pass 1: candidates and their maximum
      double v[MS];
      double mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < MS; ++k) {
        if (k < ns) {
          const unsigned sd = (unsigned)__builtin_amdgcn_readfirstlane(ldsMeta[ch * (1 + MS) + 1 + k]);
          const unsigned tb = sd >> 28;
          const double *vec = tb == 0 ? diag : (tb == 1 ? left : (tb == 2 ? down : cur));
          const bool ok = tb == 0 ? okM : (tb == 1 ? okI : (tb == 2 ? okO : active));
          const double x = vec[cS[k]];
          v[k] = ok ? x + cW[k] : -INFINITY;
          mx = dmax(mx, v[k]);
        } else v[k] = -INFINITY;
      }
      if (anyOrigin && first && origin && d == A.startNode) {   // the seed cell(0,0,start) = 0 (src/forward.defs.h:36)
        accM = 0.0; accS = 1.0f;
      } else if (first) { accM = -INFINITY; accS = 0.0f; }
      if (MODE == MB_VITERBI) {
        accM = dmax(accM, mx);
      } else {
        // pass 2: sum of exp(candidate - max) in fp32; a lone candidate gives exactly 1.0 -> log 0 -> exact result
        const double newM = dmax(accM, mx);
        const double gM = (newM == -INFINITY) ? 0.0 : newM;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < MS; ++k)
          if (k < ns) s += __builtin_amdgcn_exp2f((float)(v[k] - gM) * MED_L2E);
        if (!(first && !anyOrigin)) s += accS * __builtin_amdgcn_exp2f((float)(accM - gM) * MED_L2E);
        accS = s; accM = newM;
      }
      if (last) {
        double res;
        if (MODE == MB_VITERBI) res = accM;
        else res = ((accM == -INFINITY) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);
        if (active && d >= 0) cur[d] = res;
        if (sync) {  // wave-local: the next round reads what other lanes of this wave just wrote
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
#pragma unroll
      for (int k = 0; k < MS; ++k) { cS[k] = nS[k]; cW[k] = nW[k]; }
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
  constexpr int MS = MED_MAXSLOT;
  P.G = G; P.LPG = LPG; P.backward = backward;
  P.Spad = (S + 2) & ~1;   // >= S+1: entry [S] of every LDS vector is a -inf sentinel read by padding slots
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
  // rounds: per level, states sorted by candidate count (descending) so that a round's slot count is tight
  std::vector<Cand> tmp;
  std::vector<std::vector<int>> rounds;
  std::vector<unsigned char> sync;
  std::vector<std::array<int, 4>> degOf(S);
  for (int s = 0; s < S; ++s)
    for (int T = 0; T < 4; ++T) {
      int mx = 0;
      for (int tok = 0; tok < ntok[T]; ++tok) { cands(s, T, tok, tmp); mx = std::max(mx, (int)tmp.size()); }
      degOf[s][T] = mx;
    }
  for (int l = 0; l < nLev; ++l) {
    std::vector<int> st;
    for (int s = 0; s < S; ++s) if (lev[s] == l) st.push_back(s);
    auto tot = [&](int s) { return degOf[s][0] + degOf[s][1] + degOf[s][2] + degOf[s][3]; };
    std::stable_sort(st.begin(), st.end(), [&](int x, int y) {
      if (tot(x) != tot(y)) return tot(x) > tot(y);
      return degOf[x] > degOf[y];
    });
    for (size_t k = 0; k < st.size(); k += LPG) {
      rounds.emplace_back(st.begin() + k, st.begin() + std::min(st.size(), k + LPG));
      sync.push_back(0);
    }
    if (!sync.empty()) sync.back() = 1;
  }
  if (!sync.empty()) sync.back() = 0;
  const int R = (int)rounds.size();
  P.R = R;
  P.dest.assign((size_t)R * LPG, -1);
  for (int r = 0; r < R; ++r) for (size_t k = 0; k < rounds[r].size(); ++k) P.dest[(size_t)r * LPG + k] = (short)rounds[r][k];
  P.meta.clear(); P.src.clear(); P.eid.clear();
  for (int r = 0; r < R; ++r) {
    // unified slot list of the round: (table, j)
    std::vector<std::pair<int, int>> slots;
    for (int T = 0; T < 4; ++T) {
      int ns = 0;
      for (int s : rounds[r]) ns = std::max(ns, degOf[s][T]);
      for (int j = 0; j < ns; ++j) slots.push_back({T, j});
    }
    const int nch = std::max<int>(1, ((int)slots.size() + MS - 1) / MS);
    for (int chn = 0; chn < nch; ++chn) {
      const int k0 = chn * MS, k1 = std::min<int>((int)slots.size(), k0 + MS);
      const size_t mbase = P.meta.size();
      P.meta.resize(mbase + 1 + MS, 0);
      P.meta[mbase] = (k1 - k0) | ((chn == 0) << 4) | ((chn == nch - 1) << 5) | ((chn == nch - 1 && sync[r]) << 6) | (r << 8);
      for (int k = k0; k < k1; ++k) {
        const int T = slots[k].first, j = slots[k].second;
        const size_t b0 = P.src.size();
        P.meta[mbase + 1 + (k - k0)] = (int)((unsigned)b0 | ((unsigned)T << 28));
        P.src.resize(b0 + (size_t)ntok[T] * LPG, (uint16_t)S); P.eid.resize(b0 + (size_t)ntok[T] * LPG, 0xFFFFFFFFu);
        for (size_t lane = 0; lane < rounds[r].size(); ++lane)
          for (int tok = 0; tok < ntok[T]; ++tok) {
            cands(rounds[r][lane], T, tok, tmp);
            if (j < (int)tmp.size()) {
              P.src[b0 + (size_t)tok * LPG + lane] = tmp[j].src;
              P.eid[b0 + (size_t)tok * LPG + lane] = tmp[j].eid;
            }
          }
      }
    }
  }
  P.nChunks = (int)(P.meta.size() / (1 + MS));
}

template <class T>
static bool up(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D(program)")) return false;
  return true;
}

bool medium_refresh_weights(const mb_machine *m, MedProgram &P) {
  std::vector<double> w(P.eid.size());
  for (size_t k = 0; k < w.size(); ++k) w[k] = P.eid[k] == 0xFFFFFFFFu ? -INFINITY : m->logW[P.eid[k]];
  if (!up(P.d_w, w)) return false;
  P.dev.w = P.d_w;
  return true;
}

bool medium_build(const mb_machine *m, bool backward, int G, MedProgram &P) {
  build_program(m, backward, G, P);
  if (P.src.size() >= (1u << 28)) { set_error("machine too large for the tiled kernel family"); return false; }
  if (!up(P.d_meta, P.meta) || !up(P.d_dest, P.dest) || !up(P.d_src, P.src)) return false;
  MedProgDev &d = P.dev;
  d.S = m->S; d.Spad = P.Spad; d.R = P.R; d.LPG = P.LPG; d.G = G; d.NS = P.NS; d.nChunks = P.nChunks;
  d.nIn = m->nIn; d.nOut = m->nOut;
  d.startNode = backward ? m->S - 1 : 0; d.endNode = backward ? 0 : m->S - 1;
  d.meta = P.d_meta; d.dest = P.d_dest; d.src = P.d_src;
  return medium_refresh_weights(m, P);
}

void medium_free(MedProgram &P) {
  void *ptrs[] = {P.d_meta, P.d_dest, P.d_src, P.d_w};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  P = MedProgram();
}

// Geometry: columns per strip limited by the 160 KB LDS of a CU.
bool medium_geometry(const mb_machine *m, const MedProgram &P, MedGeom &geo) {
  const size_t perCol = (size_t)P.NS * P.Spad * sizeof(double);
  const size_t progBytes = (size_t)P.nChunks * (1 + MED_MAXSLOT) * sizeof(int) + (size_t)P.R * P.LPG * sizeof(short) + 64;
  if (progBytes > 48 * 1024) return false;
  const size_t budget = 160 * 1024 - 512 - progBytes;
  long long maxCols = (long long)(budget / perCol) - 1;   // one extra column for the halo
  if (maxCols < P.G) return false;
  int waves = (int)std::min<long long>(maxCols / P.G, 16);
  // S must be covered by 4 halo registers per thread
  while (waves < 16 && (long long)waves * 64 * 4 < m->S) ++waves;
  if ((long long)waves * 64 * 4 < m->S || (long long)waves * P.G > maxCols) return false;
  geo.waves = waves; geo.C = waves * P.G;
  geo.ldsBytes = (size_t)P.NS * (geo.C + 1) * P.Spad * sizeof(double) + progBytes;
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

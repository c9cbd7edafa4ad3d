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
//   * Forward / Backward: the silent levels are grouped into K stages, each closed transitively on the host (the same
//     construction as the tiled family, mb_medium.hip), K picked by a cost model -- one __syncthreads() per stage
//     instead of one per level;  Viterbi: level by level, one rounded add per edge, bit-identical to the reference.
//   * candidate records are streamed from L2 ([slot][lane] order: one coalesced 16-byte load per lane and slot).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "mb_wide.h"
#include "mb_device_math.h"

namespace mb {

static constexpr double W_NEG_BIG = -1e300;       // finite stand-in for -inf in the running maximum (avoids inf - inf)
static constexpr uint32_t W_IDX_MASK = 0x07ffffffu;
static constexpr uint32_t W_NO_DST = 0x07ffffffu;

// ------------------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------------------
template <int MODE, bool GV>
__global__ __launch_bounds__(1024) void k_wide_sweep(WideDev P, const PairDesc *__restrict__ pairs, const int *__restrict__ outTok,
                                                     double *__restrict__ pool, double *__restrict__ loglike, double *__restrict__ scratch) {
  extern __shared__ double wlds[];
  const PairDesc pd = pairs[blockIdx.x];
  const int tid = threadIdx.x, W = P.W, S = P.S, NV = P.NV;
  const int outLen = pd.outLen;
  double *V = GV ? scratch + (size_t)blockIdx.x * (size_t)(2 * NV + P.NX) : wlds;
  for (int k = tid; k < 2 * NV + P.NX; k += W) V[k] = -INFINITY;
  __syncthreads();
  if (tid == 0) V[S + 1] = 0.0;                     // the seed, read by the first column only
  __syncthreads();
  int prevOff = 0, curOff = NV;
  const int extraOff = 2 * NV;
  const int *out = outTok + pd.outBase;
  double *cells = pool ? pool + pd.cellBase : nullptr;
  for (int c = 0; c <= outLen; ++c) {
    const int o = P.backward ? outLen - c : c;
    const int tok = P.backward ? (o < outLen ? out[o] : 0) : (o ? out[o - 1] : 0);
    for (int r = 0; r < P.nRounds; ++r) {
      const WideRound R = P.rounds[r];
      const WideRec *rp = P.recs + (size_t)R.recBase + (size_t)tok * R.tokStride + tid;
      double m = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG;
      float s = 0.0f;
#pragma unroll 2
      for (int j = 0; j < R.depth; ++j) {
        const WideRec rc = rp[(size_t)j * W];
        const uint32_t sel = rc.src >> 30, idx = rc.src & 0x3fffffffu;
        const int base = sel == 0 ? curOff : (sel == 1 ? extraOff : prevOff);
        const double v = V[base + (int)idx] + rc.w;
        if (MODE == MB_VITERBI) m = dmax(m, v);
        else {
          const double mn = dmax(m, v);
          s = s * __expf((float)(m - mn)) + __expf((float)(v - mn));
          m = mn;
        }
      }
      const uint32_t dst = P.dsts[R.dstBase + tid];
      const int g = 1 << ((dst >> 27) & 7);
      // groups are laid out by decreasing size, so the first lane of a wavefront carries the largest group of the wavefront
      const int gWave = __builtin_amdgcn_readfirstlane(g);
      for (int k = gWave >> 1; k; k >>= 1) {
        double mo = __shfl_down(m, k, 64);
        float so = __shfl_down(s, k, 64);
        if (k >= g) { mo = (MODE == MB_VITERBI) ? -INFINITY : W_NEG_BIG; so = 0.0f; }
        if (MODE == MB_VITERBI) m = dmax(m, mo);
        else {
          const double mn = dmax(m, mo);
          s = s * __expf((float)(m - mn)) + so * __expf((float)(mo - mn));
          m = mn;
        }
      }
      if ((dst & W_IDX_MASK) != W_NO_DST) {
        const double res = (MODE == MB_VITERBI) ? m : (s > 0.0f ? m + (double)__logf(s) : -INFINITY);
        V[((dst >> 30) ? extraOff : curOff) + (int)(dst & W_IDX_MASK)] = res;
      }
      if (R.sync) __syncthreads();
    }
    // the last round of a column always synchronises: the column is complete here
    if (cells) {
      double *col = cells + (long long)o * S;
      for (int k = tid; k < S; k += W) col[k] = V[curOff + k];
    }
    if (tid == 0) V[prevOff + S + 1] = -INFINITY;   // the seed is spent (this vector is the next column's `cur`)
    const int t = prevOff; prevOff = curOff; curOff = t;
  }
  if (loglike && tid == 0) loglike[blockIdx.x] = V[prevOff + P.resultIdx];
}

// ------------------------------------------------------------------------------------------------------------
// host: program compiler
// ------------------------------------------------------------------------------------------------------------
namespace {
struct WCand { uint32_t src; double w; };
struct WNode { uint32_t dst; int stage; std::vector<std::vector<WCand>> t2; std::vector<WCand> t3; };
inline uint32_t CUR(int i) { return (uint32_t)i; }
inline uint32_t EXTRA(int i) { return (1u << 30) | (uint32_t)i; }
inline uint32_t PREV(int i) { return (2u << 30) | (uint32_t)i; }

int env_int_w(const char *name, int dflt) {
  const char *v = getenv(name);
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
struct StagePlan { int dstar; double cost; };
double plan_stage(const std::vector<const WNode *> &nodes, int nTokTables, int W, bool emit, WideProgram *P, int *nExtraDummy) {
  (void)nExtraDummy;
  const int n = (int)nodes.size();
  if (!n) return 0.0;
  std::vector<int> len(n);
  int maxLen = 1;
  bool anyT2 = false;
  for (int i = 0; i < n; ++i) {
    int t2 = 0;
    for (const auto &l : nodes[i]->t2) t2 = std::max(t2, (int)l.size());
    if (t2) anyT2 = true;
    len[i] = std::max(1, t2 + (int)nodes[i]->t3.size());
    maxLen = std::max(maxLen, len[i]);
  }
  const double cSlot = 60.0, cRound = 40.0, cShfl = 45.0, cSync = 150.0;
  auto groupOf = [&](int L, int d) { return std::min(64, pow2ceil((L + d - 1) / d)); };
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
    const int nTab = t2 ? nTokTables : 1;
    R.tokStride = t2 ? depth * W : 0;
    const WideRec padRec{-INFINITY, PREV(P->dev.S), 0};
    P->recs.resize(P->recs.size() + (size_t)nTab * depth * W, padRec);
    P->dsts.resize(P->dsts.size() + W, W_NO_DST);
    int lane = 0;
    for (int q = pos; q < e; ++q) {
      const WNode &nd = *nodes[order[q]];
      const int g = grp[order[q]];
      for (int sub = 0; sub < g; ++sub)
        P->dsts[R.dstBase + lane + sub] = (sub == 0 ? (nd.dst & 0xc0000000u) | (nd.dst & W_IDX_MASK) : W_NO_DST) | ((uint32_t)ilog2(g) << 27);
      for (int t = 0; t < nTab; ++t) {
        const std::vector<WCand> *l2 = (t < (int)nd.t2.size()) ? &nd.t2[t] : nullptr;
        const int n2 = l2 ? (int)l2->size() : 0, tot = n2 + (int)nd.t3.size();
        for (int k = 0; k < tot; ++k) {
          const WCand &cd = k < n2 ? (*l2)[k] : nd.t3[k - n2];
          const int j = k / g, sub = k % g;
          WideRec &rc = P->recs[(size_t)R.recBase + (size_t)t * R.tokStride + (size_t)j * W + lane + sub];
          rc.w = cd.w; rc.src = cd.src;
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
  (void)anyT2;
  return best;
}
}  // namespace

bool wide_applicable(const mb_machine *m) {
  if (m->nIn != 0 || m->nOut <= 0) return false;
  if (!env_int_w("MB_WIDE", 1)) return false;
  return m->S >= env_int_w("MB_WIDE_MIN_STATES", 256) && m->S < (1 << 26);
}

void wide_free(WideProgram &P) {
  if (P.d_rounds) (void)hipFree(P.d_rounds);
  if (P.d_recs) (void)hipFree(P.d_recs);
  if (P.d_dsts) (void)hipFree(P.d_dsts);
  P = WideProgram();
}

// nodes of the program for K closure stages (K = 0: levelled)
static bool wide_nodes(const mb_machine *m, bool backward, int K, long long pairCap, std::vector<WNode> &nodes, int &nExtra,
                       int &nStages, long long &nPairs) {
  const int S = m->S, nOut = m->nOut;
  const std::vector<int> &lev = backward ? m->levB : m->levF;
  const int nLev = backward ? m->nLevB : m->nLevF;
  const std::vector<int> &off = backward ? m->outOff : m->inOff;
  const std::vector<uint32_t> &perm = backward ? m->outPerm : m->inPerm;
  const int seedNode = backward ? S - 1 : 0;
  auto other = [&](uint32_t e) { return (int)(backward ? m->dst[e] : m->src[e]); };
  // candidates of a state: emitting (by output token) and silent, in the reference's iteration order
  std::vector<std::vector<std::vector<WCand>>> emitC(S, std::vector<std::vector<WCand>>(nOut + 1));
  std::vector<std::vector<std::pair<int, double>>> sil(S);
  for (int x = 0; x < S; ++x)
    for (int tok = 0; tok <= nOut; ++tok) {
      const long long rw = (long long)x * (nOut + 1) + tok;      // nIn = 0: row = (state * 1 + 0) * (nOut + 1) + tok
      for (int a = off[rw]; a < off[rw + 1]; ++a) {
        const uint32_t e = perm[a];
        const int y = other(e);
        if (tok) emitC[x][tok].push_back({PREV(y), m->logW[e]});
        else if (backward ? y > x : y < x) sil[x].push_back({y, m->logW[e]});
      }
    }
  emitC[seedNode][0].push_back({PREV(S + 1), 0.0});
  nodes.clear(); nExtra = 0; nPairs = 0;
  if (K <= 0) {
    for (int x = 0; x < S; ++x) {
      bool any = !sil[x].empty();
      for (const auto &l : emitC[x]) if (!l.empty()) any = true;
      if (!any) continue;
      WNode nd{CUR(x), lev[x], emitC[x], {}};
      for (auto &pe : sil[x]) nd.t3.push_back({CUR(pe.first), pe.second});
      nodes.push_back(std::move(nd));
    }
    nStages = nLev;
    return true;
  }
  K = std::max(1, std::min(K, std::max(1, nLev - 1)));
  std::vector<int> stg(S, 0);
  std::vector<char> isBase(S, 0);
  for (int x = 0; x < S; ++x) {
    if (lev[x] > 0) stg[x] = 1 + (int)(((long long)(lev[x] - 1) * K) / std::max(1, nLev - 1));
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
  for (int q = 0; q < S; ++q) {
    const int x = backward ? S - 1 - q : q;
    if (sil[x].empty()) continue;
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
    nPairs += (long long)touched.size();
    if (nPairs > pairCap) return false;
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

static double wide_plan(const std::vector<WNode> &nodes, int nStages, int nTok, int W, bool emit, WideProgram *P) {
  std::vector<std::vector<const WNode *>> byStage(nStages + 1);
  for (const WNode &n : nodes) byStage[std::min(n.stage, nStages)].push_back(&n);
  double cost = 0.0;
  for (auto &v : byStage) cost += plan_stage(v, nTok, W, emit, P, nullptr);
  return cost;
}

template <class T>
static bool up_w(T *&d, const std::vector<T> &h) {
  if (d) { (void)hipFree(d); d = nullptr; }
  if (!hip_ok(hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc(wide program)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(wide program)")) return false;
  return true;
}

bool wide_build(const mb_machine *m, bool backward, bool viterbi, WideProgram &P) {
  const int keepStages = P.ok ? P.stages : -1;     // a weight refresh keeps the shape that was chosen
  wide_free(P);
  P.backward = backward; P.viterbi = viterbi;
  P.W = m->S >= 768 ? 1024 : 256;
  const int S = m->S, nLev = backward ? m->nLevB : m->nLevF;
  long long nSilent = 0;
  for (long long e = 0; e < m->nTrans; ++e) nSilent += (m->inTok[e] == 0 && m->outTok[e] == 0);
  const long long pairCap = std::max<long long>(64 * (nSilent + S), 1 << 16);
  std::vector<WNode> nodes, bestNodes;
  int nExtra = 0, nStages = 0, bestExtra = 0, bestStages = 0, bestK = 0;
  long long nPairs = 0, bestPairs = 0;
  double best = 1e300;
  const bool verbose = getenv("MB_WIDE_VERBOSE") != nullptr;
  auto consider = [&](int K) {
    if (!wide_nodes(m, backward, K, pairCap, nodes, nExtra, nStages, nPairs)) return false;
    const double c = wide_plan(nodes, nStages, m->nOut + 1, P.W, false, nullptr);
    if (verbose) fprintf(stderr, "[mbhip] wide %s program, closure stages %d: modelled %.0f cycles per column (%lld pairs)\n", backward ? "backward" : "forward", K, c, nPairs);
    if (c < best) { best = c; bestNodes.swap(nodes); bestExtra = nExtra; bestStages = nStages; bestK = K; bestPairs = nPairs; }
    return true;
  };
  if (viterbi) consider(0);
  else {
    const int forced = keepStages >= 0 ? keepStages : env_int_w("MB_WIDE_CLOSURE_STAGES", -1);
    if (forced >= 0) consider(forced);
    else {
      consider(0);
      for (int K = std::max(1, nLev - 1); K >= 1; K = (K * 2) / 3) {
        if (!consider(K)) break;
        if (K == 1) break;
      }
    }
  }
  if (best >= 1e300) { set_error("wide program: no feasible shape"); return false; }
  P.stages = bestK; P.nPairs = bestPairs;
  P.NV = S + 2; P.NX = bestExtra + 1;
  P.dev.S = S;
  wide_plan(bestNodes, bestStages, m->nOut + 1, P.W, true, &P);
  if (P.rounds.empty()) { set_error("wide program: empty machine"); return false; }
  if (!up_w(P.d_rounds, P.rounds) || !up_w(P.d_recs, P.recs) || !up_w(P.d_dsts, P.dsts)) return false;
  P.dev.rounds = P.d_rounds; P.dev.recs = P.d_recs; P.dev.dsts = P.d_dsts;
  P.dev.nRounds = (int)P.rounds.size(); P.dev.NV = P.NV; P.dev.NX = P.NX; P.dev.W = P.W;
  P.dev.resultIdx = backward ? 0 : S - 1;
  P.dev.backward = backward ? 1 : 0;
  P.ok = true; P.dirty = false;
  if (verbose)
    fprintf(stderr, "[mbhip] wide %s%s program: %d stages, %zu rounds, %lld slots and %d barriers per column, %zu records, vectors %zu bytes\n",
            backward ? "backward" : "forward", viterbi ? " (max)" : "", bestK, P.rounds.size(), P.slotsPerColumn, P.nSync, P.recs.size(), P.vecBytes());
  return true;
}

static const size_t WIDE_LDS_MAX = 160 * 1024;

template <int MODE, bool GV>
static int launch_wide(const WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_out, double *pool, double *loglike,
                       double *scratch, hipStream_t st) {
  const size_t lds = GV ? 0 : P.vecBytes();
  static bool attr[2][2] = {{false, false}, {false, false}};
  if (!GV && !attr[MODE == MB_VITERBI][0]) {
    MB_HIP(hipFuncSetAttribute((const void *)k_wide_sweep<MODE, GV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_MAX));
    attr[MODE == MB_VITERBI][0] = true;
  }
  hipLaunchKernelGGL((k_wide_sweep<MODE, GV>), dim3((unsigned)nPairs), dim3(P.W), lds, st, P.dev, d_desc, d_out, pool, loglike, scratch);
  MB_HIP(hipGetLastError());
  return 0;
}

int wide_fill(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_out, double *pool,
              double *loglike, hipStream_t st) {
  (void)m;
  if (!P.ok) { set_error("wide program not built"); return 1; }
  if (nPairs <= 0) return 0;
  const bool gv = P.vecBytes() > WIDE_LDS_MAX || env_int_w("MB_WIDE_GLOBAL_VECTORS", 0);
  double *scratch = nullptr;
  if (gv) MB_HIP(hipMalloc((void **)&scratch, (size_t)nPairs * P.vecBytes()));
  int rc;
  if (P.viterbi) rc = gv ? launch_wide<MB_VITERBI, true>(P, d_desc, nPairs, d_out, pool, loglike, scratch, st)
                         : launch_wide<MB_VITERBI, false>(P, d_desc, nPairs, d_out, pool, loglike, scratch, st);
  else rc = gv ? launch_wide<MB_FORWARD, true>(P, d_desc, nPairs, d_out, pool, loglike, scratch, st)
               : launch_wide<MB_FORWARD, false>(P, d_desc, nPairs, d_out, pool, loglike, scratch, st);
  g_last_launches += 1;
  if (gv) {
    if (!rc && !hip_ok(hipStreamSynchronize(st), "wide sweep")) rc = 1;
    (void)hipFree(scratch);
  }
  return rc;
}

}  // namespace mb

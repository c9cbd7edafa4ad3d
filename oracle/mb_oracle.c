/* mb_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A plain-C, single-threaded CPU restatement of Machine Boss's DP hot path, used as the parity
 * checker for the HIP engine (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under machineboss_amd/ may import, link or call this file.
 *
 * Parity pinning: the reference C++ path cannot be compiled in this image (src/logsumexp.h:5-6 and
 * src/eval.cpp:1 include GSL, which is absent, and stand-in headers are not allowed), so this
 * restatement is pinned against
 *   - the reference's own golden vectors (tests/golden/{expect,io,machine}: copied DATA files of
 *     /root/reference/t/: fwd/back/fwdback-bitnoise-params-tiny, align-stutter-noise-difflen, 101-bitnoise-001,
 *     101-bitstutternoise-{fwd,vit}-0011, counts.json, the *_env.json envelopes),
 *   - benchmark-scale outputs of the real C++ reference recorded in SURVEY.md section 6 (tests/golden/survey_anchors.json),
 *   - outputs of the reference's second implementation of this path, its JavaScript CPU tier
 *     (js/webgpu/cpu/{forward,backward,viterbi}-2d.mjs), produced by RUNNING that code in the dev container with node:
 *     tests/golden/make_js_cases.py + make_js_goldens.mjs -> tests/golden/js/goldens.json (every Forward and Backward
 *     cell of five machines to 1e-10, Viterbi scores bit for bit; tests/test_oracle_golden.py).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference/).
 * Cell layout follows IdentityIndexMapper with a full envelope (src/dpmatrix.h:34-44,90-96):
 *   cell(inPos,outPos,state) = cells[((outPos*(inLen+1)) + inPos)*nStates + state].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* ---- log_sum_exp: src/logsumexp.h:20-26,48-90 and the table ctor src/logsumexp.cpp:8-18 ---- */
#define LSE_MAX 10
#define LSE_PREC .0001
#define LSE_ENTRIES (((int)(LSE_MAX / LSE_PREC)) + 1)

static double *lse_table = NULL;

static double lse_unary_slow(double x) { return log(1. + exp(-x)); } /* src/logsumexp.cpp:47-49 */

void mbo_init(void) {
  if (lse_table) return;
  lse_table = (double *)malloc(sizeof(double) * (LSE_ENTRIES + 1));
  for (int n = 0; n < LSE_ENTRIES; ++n) lse_table[n] = lse_unary_slow(n * LSE_PREC);
  lse_table[LSE_ENTRIES] = 0; /* never read: x >= 10 returns early */
}

static inline double lse_unary_table(double x) { /* src/logsumexp.h:48-70 */
  if (x >= LSE_MAX || isnan(x) || isinf(x)) return 0;
  if (x < 0) return -x;
  const int n = (int)(x / LSE_PREC);
  const double f0 = lse_table[n];
  const double dx = x - (n * LSE_PREC);
  const double f1 = lse_table[n + 1];
  const double df = f1 - f0;
  return f0 + df * (dx / LSE_PREC);
}

/* table variant: src/logsumexp.h:72-90 */
static inline double lse_table2(double a, double b) {
  double max, diff;
  if (a == b) { max = a; diff = 0; }
  else if (a < b) { max = b; diff = b - a; }
  else { max = a; diff = a - b; }
  return max + lse_unary_table(diff);
}

/* exact variant (-DLOG_SUM_EXP_SLOW build of the reference): src/logsumexp.h:50-52 with the same
 * a==b / max / diff prologue; (-inf,-inf) -> -inf + log(2) = -inf, (x,-inf) -> x + log(1+0) = x. */
static inline double lse_exact2(double a, double b) {
  double max, diff;
  if (a == b) { max = a; diff = 0; }
  else if (a < b) { max = b; diff = b - a; }
  else { max = a; diff = a - b; }
  return max + lse_unary_slow(diff);
}

double mbo_log_sum_exp(double a, double b, int exact) { mbo_init(); return exact ? lse_exact2(a, b) : lse_table2(a, b); }

/* ---- flattened evaluated machine ---------------------------------------------------------- */
typedef struct {
  int nStates, nInTok, nOutTok; /* alphabet sizes, excluding epsilon (token 0) */
  long nTrans;
  uint32_t *src, *dst;
  uint16_t *inTok, *outTok;
  double *logW;
  long K;             /* (nInTok+1)*(nOutTok+1) label keys per state */
  long *inOff;        /* [nStates*K + 1] CSR by (dst, inTok, outTok) into inEdge  */
  uint32_t *inEdge;   /* edge ids in the reference's `incoming` iteration order    */
  long *outOff;       /* [nStates*K + 1] CSR by (src, inTok, outTok) into outEdge */
  uint32_t *outEdge;  /* edge ids in the reference's `outgoing` iteration order    */
} mbo_machine;

/* Build both CSR views.  Edges arrive in global order e = transOffset[src] + transIndex, i.e. the order
 * EvaluatedMachine::init inserts them (src/eval.cpp:47-69).  The nested map<in,map<out,multimap<state,..>>>
 * (src/eval.h:66-68) then iterates (in, out, state, insertion order); a counting sort by key followed by a
 * stable sort on the other endpoint reproduces that exactly. */
static void build_csr(const mbo_machine *m, int incoming, long *off, uint32_t *edge) {
  const long nKeys = (long)m->nStates * m->K;
  memset(off, 0, sizeof(long) * (nKeys + 1));
  for (long e = 0; e < m->nTrans; ++e) {
    const long st = incoming ? m->dst[e] : m->src[e];
    const long key = (st * (m->nInTok + 1) + m->inTok[e]) * (m->nOutTok + 1) + m->outTok[e];
    off[key + 1]++;
  }
  for (long k = 0; k < nKeys; ++k) off[k + 1] += off[k];
  long *fill = (long *)malloc(sizeof(long) * nKeys);
  memcpy(fill, off, sizeof(long) * nKeys);
  for (long e = 0; e < m->nTrans; ++e) { /* ascending e == ascending (src, transIndex) */
    const long st = incoming ? m->dst[e] : m->src[e];
    const long key = (st * (m->nInTok + 1) + m->inTok[e]) * (m->nOutTok + 1) + m->outTok[e];
    edge[fill[key]++] = (uint32_t)e;
  }
  free(fill);
  /* within a key: order by the other endpoint, stable (insertion sort; lists are short) */
  for (long k = 0; k < nKeys; ++k)
    for (long a = off[k] + 1; a < off[k + 1]; ++a) {
      const uint32_t e = edge[a];
      const uint32_t ke = incoming ? m->src[e] : m->dst[e];
      long b = a - 1;
      while (b >= off[k] && (incoming ? m->src[edge[b]] : m->dst[edge[b]]) > ke) { edge[b + 1] = edge[b]; --b; }
      edge[b + 1] = e;
    }
}

mbo_machine *mbo_machine_create(int nStates, int nInTok, int nOutTok, long nTrans, const uint32_t *src,
                                const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok,
                                const double *logW) {
  mbo_init();
  mbo_machine *m = (mbo_machine *)calloc(1, sizeof(mbo_machine));
  m->nStates = nStates; m->nInTok = nInTok; m->nOutTok = nOutTok; m->nTrans = nTrans;
  m->K = (long)(nInTok + 1) * (nOutTok + 1);
  m->src = (uint32_t *)malloc(sizeof(uint32_t) * (nTrans + 1)); memcpy(m->src, src, sizeof(uint32_t) * nTrans);
  m->dst = (uint32_t *)malloc(sizeof(uint32_t) * (nTrans + 1)); memcpy(m->dst, dst, sizeof(uint32_t) * nTrans);
  m->inTok = (uint16_t *)malloc(sizeof(uint16_t) * (nTrans + 1)); memcpy(m->inTok, inTok, sizeof(uint16_t) * nTrans);
  m->outTok = (uint16_t *)malloc(sizeof(uint16_t) * (nTrans + 1)); memcpy(m->outTok, outTok, sizeof(uint16_t) * nTrans);
  m->logW = (double *)malloc(sizeof(double) * (nTrans + 1)); memcpy(m->logW, logW, sizeof(double) * nTrans);
  const long nKeys = (long)nStates * m->K;
  m->inOff = (long *)malloc(sizeof(long) * (nKeys + 1));
  m->outOff = (long *)malloc(sizeof(long) * (nKeys + 1));
  m->inEdge = (uint32_t *)malloc(sizeof(uint32_t) * (nTrans + 1));
  m->outEdge = (uint32_t *)malloc(sizeof(uint32_t) * (nTrans + 1));
  build_csr(m, 1, m->inOff, m->inEdge);
  build_csr(m, 0, m->outOff, m->outEdge);
  return m;
}

void mbo_machine_set_weights(mbo_machine *m, const double *logW) { memcpy(m->logW, logW, sizeof(double) * m->nTrans); }

void mbo_machine_destroy(mbo_machine *m) {
  if (!m) return;
  free(m->src); free(m->dst); free(m->inTok); free(m->outTok); free(m->logW);
  free(m->inOff); free(m->inEdge); free(m->outOff); free(m->outEdge); free(m);
}

void mbo_incoming_order(const mbo_machine *m, uint32_t *out) { memcpy(out, m->inEdge, sizeof(uint32_t) * m->nTrans); }
void mbo_outgoing_order(const mbo_machine *m, uint32_t *out) { memcpy(out, m->outEdge, sizeof(uint32_t) * m->nTrans); }

#define KEY(m, st, it, ot) ((((long)(st)) * ((m)->nInTok + 1) + (it)) * ((m)->nOutTok + 1) + (ot))
#define CELL(cells, I, S, i, o, s) ((cells)[(((long)(o)) * (I) + (i)) * (S) + (s)])

enum { MBO_SUM_TABLE = 0, MBO_SUM_EXACT = 1, MBO_MAX = 2 };

/* Envelope (src/seqpair.h:75-97): (x,y) is inside <=> inStart[y] <= x < inEnd[y].  NULL = full envelope.  The fills
 * below visit only the cells inside (src/forward.defs.h:29, viterbi.cpp:23, backward.cpp:25); every other cell keeps the
 * -inf that alloc() stored (src/dpmatrix.defs.h:36), which is also what the const cell() accessor returns for a
 * position outside the envelope (src/dpmatrix.h:142-144).  The matrix here is always the FULL rectangle. */
static const long *env_start = NULL, *env_end = NULL;
void mbo_set_envelope(const long *inStart, const long *inEnd) { env_start = inStart; env_end = inEnd; }
#define ENV_LO(o) (env_start ? env_start[o] : 0)
#define ENV_HI(o, inLen) (env_end ? env_end[o] : (inLen) + 1)

static inline double reduce2(int mode, double a, double b) {
  if (mode == MBO_MAX) return a > b ? a : (b > a ? b : a); /* std::max(a,b): returns a unless a<b (src/dpmatrix.h:122) */
  return mode == MBO_SUM_EXACT ? lse_exact2(a, b) : lse_table2(a, b);
}

/* DPMatrix::accumulate over `incoming` (src/dpmatrix.h:101-115): fold cell(srcPos, edge.src) + logWeight */
static inline double acc_in(const mbo_machine *m, int mode, double ll, int d, int it, int ot, const double *srcCell) {
  const long k = KEY(m, d, it, ot);
  for (long a = m->inOff[k]; a < m->inOff[k + 1]; ++a) {
    const uint32_t e = m->inEdge[a];
    ll = reduce2(mode, ll, srcCell[m->src[e]] + m->logW[e]);
  }
  return ll;
}

static inline double acc_out(const mbo_machine *m, int mode, double ll, int s, int it, int ot, const double *dstCell) {
  const long k = KEY(m, s, it, ot);
  for (long a = m->outOff[k]; a < m->outOff[k + 1]; ++a) {
    const uint32_t e = m->outEdge[a];
    ll = reduce2(mode, ll, dstCell[m->dst[e]] + m->logW[e]);
  }
  return ll;
}

/* MappedForwardMatrix::fill (src/forward.defs.h:23-49) with mode = table/exact sum;
 * ViterbiMatrix::fill (src/viterbi.cpp:18-43) with mode = MBO_MAX and startState = 0.
 * cells must hold (inLen+1)*(outLen+1)*nStates doubles; every cell is written once. */
void mbo_fill_forward(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                      int mode, int startState, double *cells) {
  const long I = inLen + 1; const int S = m->nStates;
  for (long x = 0; x < I * (outLen + 1) * S; ++x) cells[x] = NEG_INF; /* alloc() fills with -inf (src/dpmatrix.defs.h:36) */
  for (long o = 0; o <= outLen; ++o) {
    const int ot = o ? out[o - 1] : 0;
    for (long i = ENV_LO(o); i < ENV_HI(o, inLen); ++i) {
      const int it = i ? in[i - 1] : 0;
      double *cur = &CELL(cells, I, S, i, o, 0);
      for (int d = 0; d < S; ++d) {
        double ll = (i || o || d != startState) ? NEG_INF : 0;
        if (i && o) ll = acc_in(m, mode, ll, d, it, ot, &CELL(cells, I, S, i - 1, o - 1, 0));
        if (i) ll = acc_in(m, mode, ll, d, it, 0, &CELL(cells, I, S, i - 1, o, 0));
        if (o) ll = acc_in(m, mode, ll, d, 0, ot, &CELL(cells, I, S, i, o - 1, 0));
        ll = acc_in(m, mode, ll, d, 0, 0, cur);
        cur[d] = ll;
      }
    }
  }
}

/* RollingOutputForwardMatrix (src/dpmatrix.h:46-58, target/boss.cpp:799): two rows of (inLen+1)*nStates.
 * Returns logLike() = cell(inLen,outLen,endState) (src/forward.defs.h:51-55). */
double mbo_forward_loglike(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                           int mode) {
  const long I = inLen + 1; const int S = m->nStates;
  double *rows = (double *)malloc(sizeof(double) * 2 * I * S);
  for (long x = 0; x < 2 * I * S; ++x) rows[x] = NEG_INF;
  for (long o = 0; o <= outLen; ++o) {
    const int ot = o ? out[o - 1] : 0;
    double *row = rows + (o % 2) * I * S, *prev = rows + ((o + 1) % 2) * I * S;
    for (long i = 0; i <= inLen; ++i) {
      const int it = i ? in[i - 1] : 0;
      double *cur = row + i * S;
      /* NB the reference does not clear the recycled row; cells are overwritten in state order and a state only
       * reads lower-numbered states of its own supercell through silent edges (advancing machine), except a
       * silent self-loop on state 0 which would see the stale value.  We clear, matching the full matrix. */
      for (int d = 0; d < S; ++d) cur[d] = NEG_INF;
      for (int d = 0; d < S; ++d) {
        double ll = (i || o || d != 0) ? NEG_INF : 0;
        if (i && o) ll = acc_in(m, mode, ll, d, it, ot, prev + (i - 1) * S);
        if (i) ll = acc_in(m, mode, ll, d, it, 0, row + (i - 1) * S);
        if (o) ll = acc_in(m, mode, ll, d, 0, ot, prev + i * S);
        ll = acc_in(m, mode, ll, d, 0, 0, cur);
        cur[d] = ll;
      }
    }
  }
  const double r = rows[(outLen % 2) * I * S + inLen * S + (S - 1)];
  free(rows);
  return r;
}

/* BackwardMatrix::fill (src/backward.cpp:18-46) */
void mbo_fill_backward(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                       int mode, double *cells) {
  const long I = inLen + 1; const int S = m->nStates;
  for (long x = 0; x < I * (outLen + 1) * S; ++x) cells[x] = NEG_INF;
  for (long o = outLen; o >= 0; --o) {
    const int endOfOutput = (o == outLen);
    const int ot = endOfOutput ? 0 : out[o];
    for (long i = ENV_HI(o, inLen) - 1; i >= ENV_LO(o); --i) {
      const int endOfInput = (i == inLen);
      const int it = endOfInput ? 0 : in[i];
      double *cur = &CELL(cells, I, S, i, o, 0);
      for (int s = S - 1; s >= 0; --s) {
        double ll = (endOfInput && endOfOutput && s == S - 1) ? 0 : NEG_INF;
        if (!endOfInput && !endOfOutput) ll = acc_out(m, mode, ll, s, it, ot, &CELL(cells, I, S, i + 1, o + 1, 0));
        if (!endOfInput) ll = acc_out(m, mode, ll, s, it, 0, &CELL(cells, I, S, i + 1, o, 0));
        if (!endOfOutput) ll = acc_out(m, mode, ll, s, 0, ot, &CELL(cells, I, S, i, o + 1, 0));
        ll = acc_out(m, mode, ll, s, 0, 0, cur);
        cur[s] = ll;
      }
    }
  }
}

/* BackwardMatrix::getCounts + accumulateCounts + transitionCounter
 * (src/backward.cpp:58-87, src/backward.h:12-18,37-42).  counts[e] is indexed by global edge id
 * e = transOffset[src] + transIndex, i.e. MachineCounts::count[src][transIndex] flattened (src/counts.cpp:45-50). */
static inline void cnt_out(const mbo_machine *m, double logOdds, int s, int it, int ot, const double *bwdDst,
                           double *counts) {
  const long k = KEY(m, s, it, ot);
  for (long a = m->outOff[k]; a < m->outOff[k + 1]; ++a) {
    const uint32_t e = m->outEdge[a];
    const double tll = bwdDst[m->dst[e]] + m->logW[e];
    counts[e] += exp(logOdds + tll);
  }
}

void mbo_get_counts(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                    const double *fwd, const double *bwd, double *counts) {
  const long I = inLen + 1; const int S = m->nStates;
  const double ll = CELL(bwd, I, S, 0, 0, 0); /* BackwardMatrix::logLike (src/backward.cpp:48-50) */
  for (long o = outLen; o >= 0; --o) {
    const int endOfOutput = (o == outLen);
    const int ot = endOfOutput ? 0 : out[o];
    for (long i = ENV_HI(o, inLen) - 1; i >= ENV_LO(o); --i) {
      const int endOfInput = (i == inLen);
      const int it = endOfInput ? 0 : in[i];
      for (int s = S - 1; s >= 0; --s) {
        const double logOdds = CELL(fwd, I, S, i, o, s) - ll;
        if (!endOfInput && !endOfOutput) cnt_out(m, logOdds, s, it, ot, &CELL(bwd, I, S, i + 1, o + 1, 0), counts);
        if (!endOfInput) cnt_out(m, logOdds, s, it, 0, &CELL(bwd, I, S, i + 1, o, 0), counts);
        if (!endOfOutput) cnt_out(m, logOdds, s, 0, ot, &CELL(bwd, I, S, i, o + 1, 0), counts);
        cnt_out(m, logOdds, s, 0, 0, &CELL(bwd, I, S, i, o, 0), counts);
      }
    }
  }
}

/* MachineCounts::add (src/counts.cpp:57-64): Forward + Backward + getCounts; returns forward.logLike().
 * counts accumulates (+=), as MachineCounts does over a SeqPairList (src/counts.cpp:37-43). */
double mbo_counts_add(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                      int mode, double *counts) {
  const long n = (inLen + 1) * (outLen + 1) * (long)m->nStates;
  double *fwd = (double *)malloc(sizeof(double) * n), *bwd = (double *)malloc(sizeof(double) * n);
  mbo_fill_forward(m, in, inLen, out, outLen, mode, 0, fwd);
  mbo_fill_backward(m, in, inLen, out, outLen, mode, bwd);
  mbo_get_counts(m, in, inLen, out, outLen, fwd, bwd, counts);
  const double ll = fwd[n - 1];
  free(fwd); free(bwd);
  return ll;
}

/* DPMatrix::traceBack with selectMaxTrans (src/dpmatrix.defs.h:61-110,171-174).
 * Walks from (inLen,outLen,endState) to (0,0,state 0) over a filled matrix; at every step rebuilds the
 * candidate list in the order match / in-only / out-only / silent (:93-99) and takes the FIRST maximum
 * (std::max_element).  Writes global edge ids start->end into path[]; returns the number of transitions,
 * -1 if the end cell is -inf ("Can't do traceback", :84), -2 if pathCap is too small. */
long mbo_traceback(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen,
                   const double *cells, uint32_t *path, long pathCap) {
  const long I = inLen + 1; const int S = m->nStates;
  long i = inLen, o = outLen; int s = S - 1;
  if (!(CELL(cells, I, S, i, o, s) > NEG_INF)) return -1;
  long n = 0;
  while (i > 0 || o > 0 || s != 0) {
    const int it = i ? in[i - 1] : 0;
    const int ot = o ? out[o - 1] : 0;
    double best = 0; long bestE = -1;
    for (int grp = 0; grp < 4; ++grp) {
      long k; const double *sc;
      if (grp == 0) { if (!(i && o)) continue; k = KEY(m, s, it, ot); sc = &CELL(cells, I, S, i - 1, o - 1, 0); }
      else if (grp == 1) { if (!i) continue; k = KEY(m, s, it, 0); sc = &CELL(cells, I, S, i - 1, o, 0); }
      else if (grp == 2) { if (!o) continue; k = KEY(m, s, 0, ot); sc = &CELL(cells, I, S, i, o - 1, 0); }
      else { k = KEY(m, s, 0, 0); sc = &CELL(cells, I, S, i, o, 0); }
      for (long a = m->inOff[k]; a < m->inOff[k + 1]; ++a) {
        const uint32_t e = m->inEdge[a];
        const double v = sc[m->src[e]] + m->logW[e];
        if (bestE < 0 || best < v) { best = v; bestE = e; } /* max_element: first element not less than any other */
      }
    }
    if (bestE < 0) return -3; /* empty candidate list: the reference would index an empty vector */
    if (n >= pathCap) return -2;
    path[n++] = (uint32_t)bestE;
    if (m->inTok[bestE]) --i;
    if (m->outTok[bestE]) --o;
    s = (int)m->src[bestE];
  }
  for (long a = 0, b = n - 1; a < b; ++a, --b) { const uint32_t t = path[a]; path[a] = path[b]; path[b] = t; }
  return n;
}

/* ---- std::mt19937 as the reference's walkers consume it ------------------------------------------------------------
 * DPMatrix::randomTransSelector (src/dpmatrix.defs.h:176-186) draws through random_index / random_double
 * (src/util.h:102-106,151-165): one 32-bit output of std::mt19937 per choice, divided by 2^32.  The generator below is
 * the published MT19937 algorithm (Matsumoto & Nishimura 1998: init_genrand / genrand_int32), which is what
 * std::mt19937(seed) is defined to be. */
typedef struct { uint32_t mt[624]; int idx; } mbo_mt19937;

void mbo_mt_seed(mbo_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t mbo_mt_next(mbo_mt19937 *g) {
  if (g->idx >= 624) {
    for (int k = 0; k < 624; ++k) {
      const uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
      g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}

mbo_mt19937 *mbo_mt_create(uint32_t seed) { mbo_mt19937 *g = (mbo_mt19937 *)malloc(sizeof(mbo_mt19937)); mbo_mt_seed(g, seed); return g; }
void mbo_mt_destroy(mbo_mt19937 *g) { free(g); }

/* random_double (src/util.h:102-106) divides one generator output by numeric_limits<Generator::result_type>::max() + 1.
 * For std::mt19937 the result_type is uint_fast32_t, which libstdc++ on LP64 Linux -- the platform the reference is built
 * on here and on the GPU box -- makes a 64-BIT unsigned long: the divisor is 2^64 although the generator only ever
 * returns 32 bits, so the "uniform" variate lies in [0, 2^-32) and random_index all but always takes the first candidate
 * of non-negligible weight.  That is what the reference does on this platform, so it is what the restatement does by
 * default (SURVEY.md section 9 does not list it: quirk Q12 in DESIGN.md); with libc++ (macOS) uint_fast32_t is 32 bits
 * and the divisor 2^32 -- mbo_set_result_bits(32) selects that reading. */
static double mt_denominator = 18446744073709551616.0;   /* ((double) ULONG_MAX) + 1 */
void mbo_set_result_bits(int bits) { mt_denominator = bits == 32 ? 4294967296.0 : 18446744073709551616.0; }

/* random_index over exp(logWeights) (src/util.h:151-165 via src/dpmatrix.defs.h:178-184); -1 on zero total weight */
static long select_random(const double *ll, long n, mbo_mt19937 *g) {
  double norm = 0;
  for (long k = 0; k < n; ++k) norm += exp(ll[k]);
  if (!(norm > 0)) return -1;
  double variate = (mbo_mt_next(g) / mt_denominator) * norm;
  for (long k = 0; k < n; ++k)
    if ((variate -= exp(ll[k])) <= 0) return k;
  return n;
}

static long select_max(const double *ll, long n) { /* selectMaxTrans: std::max_element, first maximum (src/dpmatrix.defs.h:171-174) */
  long best = 0;
  for (long k = 1; k < n; ++k) if (ll[best] < ll[k]) best = k;
  return best;
}

/* the TraceTerminator the tests use is Machine::downsample's (src/machine.cpp:2057-2064): a transition seen before stops
 * the trace, a new one is marked and the trace goes on.  mask == NULL: never stop. */
static int stop_mask(uint8_t *mask, uint32_t e) {
  if (!mask) return 0;
  if (mask[e]) return 1;
  mask[e] = 1;
  return 0;
}

/* DPMatrix::traceBack (m, inPos, outPos, s, stopTrace, selectTrans), src/dpmatrix.defs.h:82-110.
 * selector 0 = selectMaxTrans, 1 = randomTransSelector(rng).  edges[] receives the global edge id of every step in the
 * order the steps are taken (end -> start); the step that makes stopTrace return true is included.
 * Returns the number of steps, -1: start cell is -inf, -2: cap too small, -3: empty candidate list / zero weights. */
long mbo_trace_back(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen, const double *cells,
                    long i, long o, int s, int selector, mbo_mt19937 *rng, uint8_t *mask, uint32_t *edges, long cap) {
  const long I = inLen + 1; const int S = m->nStates;
  if (i < 0 || i > inLen || o < 0 || o > outLen) return -1;   /* the const cell() reads -inf outside the envelope (src/dpmatrix.h:142-144) */
  if (!(CELL(cells, I, S, i, o, s) > NEG_INF)) return -1;
  double *ll = (double *)malloc(sizeof(double) * (m->nTrans + 1));
  uint32_t *cand = (uint32_t *)malloc(sizeof(uint32_t) * (m->nTrans + 1));
  long n = 0;
  while (i > 0 || o > 0 || s != 0) {
    const int it = i ? in[i - 1] : 0;
    const int ot = o ? out[o - 1] : 0;
    long nc = 0;
    for (int grp = 0; grp < 4; ++grp) {
      long k; const double *sc;
      if (grp == 0) { if (!(i && o)) continue; k = KEY(m, s, it, ot); sc = &CELL(cells, I, S, i - 1, o - 1, 0); }
      else if (grp == 1) { if (!i) continue; k = KEY(m, s, it, 0); sc = &CELL(cells, I, S, i - 1, o, 0); }
      else if (grp == 2) { if (!o) continue; k = KEY(m, s, 0, ot); sc = &CELL(cells, I, S, i, o - 1, 0); }
      else { k = KEY(m, s, 0, 0); sc = &CELL(cells, I, S, i, o, 0); }
      for (long a = m->inOff[k]; a < m->inOff[k + 1]; ++a) {
        const uint32_t e = m->inEdge[a];
        cand[nc] = e; ll[nc++] = sc[m->src[e]] + m->logW[e];
      }
    }
    if (!nc) { n = -3; break; }
    const long best = selector ? select_random(ll, nc, rng) : select_max(ll, nc);
    if (best < 0 || best >= nc) { n = -3; break; }
    const uint32_t e = cand[best];
    if (n >= cap) { n = -2; break; }
    edges[n++] = e;
    if (m->inTok[e]) --i;
    if (m->outTok[e]) --o;
    s = (int)m->src[e];
    if (stop_mask(mask, e)) break;
  }
  free(ll); free(cand);
  return n;
}

/* DPMatrix::traceForward (m, inPos, outPos, s, stopTrace, selectTrans), src/dpmatrix.defs.h:128-159, over a Backward
 * matrix.  Same conventions; stopTrace is asked BEFORE the move (:149), and the step it stops at is not recorded
 * unless the terminator itself records it -- here the mask terminator marks it, and it IS listed (as Machine::downsample
 * counts it). */
long mbo_trace_forward(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen, const double *cells,
                       long i, long o, int s, int selector, mbo_mt19937 *rng, uint8_t *mask, uint32_t *edges, long cap) {
  const long I = inLen + 1; const int S = m->nStates;
  if (i < 0 || i > inLen || o < 0 || o > outLen) return -1;
  if (!(CELL(cells, I, S, i, o, s) > NEG_INF)) return -1;
  double *ll = (double *)malloc(sizeof(double) * (m->nTrans + 1));
  uint32_t *cand = (uint32_t *)malloc(sizeof(uint32_t) * (m->nTrans + 1));
  long n = 0;
  while (i < inLen || o < outLen || s != S - 1) {
    const int endIn = (i == inLen), endOut = (o == outLen);
    const int it = endIn ? 0 : in[i];
    const int ot = endOut ? 0 : out[o];
    long nc = 0;
    for (int grp = 0; grp < 4; ++grp) {
      long k; const double *dc;
      if (grp == 0) { if (endIn || endOut) continue; k = KEY(m, s, it, ot); dc = &CELL(cells, I, S, i + 1, o + 1, 0); }
      else if (grp == 1) { if (endIn) continue; k = KEY(m, s, it, 0); dc = &CELL(cells, I, S, i + 1, o, 0); }
      else if (grp == 2) { if (endOut) continue; k = KEY(m, s, 0, ot); dc = &CELL(cells, I, S, i, o + 1, 0); }
      else { k = KEY(m, s, 0, 0); dc = &CELL(cells, I, S, i, o, 0); }
      for (long a = m->outOff[k]; a < m->outOff[k + 1]; ++a) {
        const uint32_t e = m->outEdge[a];
        cand[nc] = e; ll[nc++] = dc[m->dst[e]] + m->logW[e];
      }
    }
    if (!nc) { n = -3; break; }
    const long best = selector ? select_random(ll, nc, rng) : select_max(ll, nc);
    if (best < 0 || best >= nc) { n = -3; break; }
    const uint32_t e = cand[best];
    if (n >= cap) { n = -2; break; }
    edges[n++] = e;
    if (stop_mask(mask, e)) break;
    if (m->inTok[e]) ++i;
    if (m->outTok[e]) ++o;
    s = (int)m->dst[e];
  }
  free(ll); free(cand);
  return n;
}

/* BackwardMatrix::getCounts with a visitor (src/backward.cpp:58-87): every (cell, outgoing transition) usage in visit
 * order, as BackwardMatrix::transitionSorter sees them (src/backward.h:28-34): position = the DESTINATION cell of the
 * transition (accumulateCounts is handed inPos+1 / outPos+1, src/backward.cpp:77-83).  Returns the number of usages. */
long mbo_post_trans(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen, const double *fwd,
                    const double *bwd, long *pInPos, long *pOutPos, uint32_t *pEdge, double *pWeight, long cap) {
  const long I = inLen + 1; const int S = m->nStates;
  const double ll = CELL(bwd, I, S, 0, 0, 0);
  long n = 0;
  for (long o = outLen; o >= 0; --o) {
    const int endOut = (o == outLen);
    const int ot = endOut ? 0 : out[o];
    for (long i = ENV_HI(o, inLen) - 1; i >= ENV_LO(o); --i) {
      const int endIn = (i == inLen);
      const int it = endIn ? 0 : in[i];
      for (int s = S - 1; s >= 0; --s) {
        const double logOdds = CELL(fwd, I, S, i, o, s) - ll;
        for (int grp = 0; grp < 4; ++grp) {
          long k, di, dq;
          if (grp == 0) { if (endIn || endOut) continue; k = KEY(m, s, it, ot); di = i + 1; dq = o + 1; }
          else if (grp == 1) { if (endIn) continue; k = KEY(m, s, it, 0); di = i + 1; dq = o; }
          else if (grp == 2) { if (endOut) continue; k = KEY(m, s, 0, ot); di = i; dq = o + 1; }
          else { k = KEY(m, s, 0, 0); di = i; dq = o; }
          for (long a = m->outOff[k]; a < m->outOff[k + 1]; ++a) {
            const uint32_t e = m->outEdge[a];
            if (n >= cap) return -2;
            pInPos[n] = di; pOutPos[n] = dq; pEdge[n] = e;
            const double tll = CELL(bwd, I, S, di, dq, m->dst[e]) + m->logW[e];   /* DPMatrix::iterate forms cell + logWeight first (src/dpmatrix.h:111) */
            pWeight[n++] = exp(logOdds + tll);                                      /* accumulateCounts, src/backward.h:38-40 */
          }
        }
      }
    }
  }
  return n;
}

/* BackwardMatrix::traceFrom with a terminator (src/backward.cpp:99-108): the transition itself, then a traceback over
 * the Forward matrix from (inPos,outPos,src), then a traceforward over the Backward matrix from the transition's far end.
 * edges[]: the transition first (if the terminator did not stop on it: it is listed either way, as the terminator has
 * marked it), then the traceback's steps, then the traceforward's. */
long mbo_trace_from(const mbo_machine *m, const int32_t *in, long inLen, const int32_t *out, long outLen, const double *fwd,
                    const double *bwd, long i, long o, uint32_t e, uint8_t *mask, uint32_t *edges, long cap) {
  long n = 0;
  if (cap < 1) return -2;
  edges[n++] = e;
  if (stop_mask(mask, e)) return n;
  const int s = (int)m->src[e];
  long k = mbo_trace_back(m, in, inLen, out, outLen, fwd, i, o, s, 0, NULL, mask, edges + n, cap - n);
  if (k < 0) return k;
  n += k;
  const long ni = i + (m->inTok[e] ? 1 : 0), no = o + (m->outTok[e] ? 1 : 0);
  k = mbo_trace_forward(m, in, inLen, out, outLen, bwd, ni, no, (int)m->dst[e], 0, NULL, mask, edges + n, cap - n);
  if (k < 0) return k;
  return n + k;
}

// mb_api.hip -- implementation of the C-ABI declared in include/mbhip.h.
//
// Host orchestration only: device memory pools, chunking of a batch so that materialised matrices fit the HBM
// budget (288 GB per MI355X), kernel-family selection, HIP-event timing on the library stream.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mb_internal.h"
#include "mb_jit.h"
#include "mb_medium.h"
#include "mb_small.h"
#include "mb_usage.h"
#include "mb_wide.h"
#include "mb_wide_jit.h"

namespace mb {

thread_local std::string g_err;
thread_local double g_last_ms = 0.0;
thread_local const char *g_last_kernel = "";
thread_local long long g_last_launches = 0;
hipStream_t g_stream = nullptr;
int g_kernel_choice = 0;
size_t g_mem_budget = 0;
static bool g_init = false;

void set_error(const std::string &msg) { g_err = msg; }

// the library's own option table (mb_set_option); readers run inside API calls, under the API lock
static std::map<std::string, std::string> g_options;
const char *opt_env(const char *name) {
  if (!g_options.empty()) {
    auto it = g_options.find(name);
    if (it != g_options.end()) return it->second.c_str();
  }
  return getenv(name);
}
void opt_set(const char *name, const char *value) {
  if (value && *value) g_options[name] = value; else g_options.erase(name);
}

bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

// One lock for every entry point that touches the process-wide state (stream, workspaces, compiled programs): the reference
// has no threads on this path and the library is built for one host thread per process, but two threads calling in
// must not corrupt the pools.  The outermost entry also un-pins the workspaces of the previous call.
static std::recursive_mutex g_api_mutex;
static int g_api_depth = 0;
static void ws_begin_call();
struct ApiGuard {
  std::lock_guard<std::recursive_mutex> lk;
  ApiGuard() : lk(g_api_mutex) { if (g_api_depth++ == 0) ws_begin_call(); }
  ~ApiGuard() { --g_api_depth; }
};
// the same lock without the "a new call begins" bookkeeping: option and introspection entry points (they touch the environment /
// the planners' process-wide state, which guarded calls of other threads read)
struct ApiLock {
  std::lock_guard<std::recursive_mutex> lk;
  ApiLock() : lk(g_api_mutex) {}
};
bool g_deterministic = false;
static int g_device = -1;   // device the library's stream and workspaces live on

static int ensure_init() {
  if (g_init) return 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    set_error("no HIP device visible: the Machine Boss DP engine has no CPU fallback");
    return 1;
  }
  MB_HIP(hipGetDevice(&g_device));
  MB_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  g_init = true;
  return 0;
}

// Grow-only device workspaces, kept across calls so that steady-state batch calls do no hipMalloc/hipFree
// (a 200 GB hipMalloc costs far more than the kernels it feeds).  Slot 0/1: matrix pools, 2: halo columns.
struct Workspace { void *p = nullptr; size_t bytes = 0; bool pinned = false; };
static const int WS_SLOTS = 16;
static Workspace g_ws[WS_SLOTS];   // 0/1 matrix pools, 2 halo columns, 3..7 Viterbi path buffers, 8.. small-machine family

static size_t cached_bytes() { size_t t = 0; for (const Workspace &w : g_ws) t += w.bytes; return t; }

// A slot handed out during the current API call is pinned until the next call begins; growing one slot may release every
// unpinned one (a Viterbi batch that needs 80 % of HBM in slot 0 must be able to reclaim the Backward pool a previous
// count sweep left in slot 1 -- budget_bytes() counts cached bytes as available).
static void ws_begin_call() { for (Workspace &w : g_ws) w.pinned = false; }   // (declared above, next to ApiGuard)
// what the pools cost this process so far (mb_alloc_stats): a 200 GB hipMalloc / hipFree pair takes SECONDS, so a caller -- and the
// tests -- must be able to see that steady-state calls do none
struct AllocStats { long long allocs = 0, frees = 0, evictions = 0; unsigned long long bytes = 0; double ms = 0.0; };
static AllocStats g_alloc;
static void ws_free_slot(Workspace &w) {
  if (!w.p) return;
  const auto t0 = std::chrono::steady_clock::now();
  (void)hipFree(w.p);
  g_alloc.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ++g_alloc.frees;
  w.p = nullptr; w.bytes = 0;
}
static void ws_release_unpinned(int except) {
  for (int k = 0; k < WS_SLOTS; ++k) {
    Workspace &w = g_ws[k];
    if (k == except || w.pinned || !w.p) continue;
    ws_free_slot(w); ++g_alloc.evictions;
  }
}

size_t ws_bytes(int slot) { return g_ws[slot].p ? g_ws[slot].bytes : 0; }

void *ws_get(int slot, size_t bytes) {
  Workspace &w = g_ws[slot];
  w.pinned = true;
  if (w.bytes >= bytes && w.p) return w.p;
  ws_free_slot(w);
  const size_t want = std::max<size_t>(bytes, 256);
  size_t freeB = 0, totalB = 0;
  // an explicit budget (mb_set_memory_budget) bounds everything the library keeps; otherwise only a shortage evicts
  const bool over = g_mem_budget && cached_bytes() + want > g_mem_budget;
  if (over || (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB < want + ((size_t)256 << 20))) ws_release_unpinned(slot);
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&w.p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ws_release_unpinned(slot);
    e = hipMalloc(&w.p, want);
  }
  g_alloc.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!hip_ok(e, "hipMalloc(workspace)")) { w.p = nullptr; return nullptr; }
  ++g_alloc.allocs; g_alloc.bytes += want;
  w.bytes = want;
  return w.p;
}

// Before a call takes SEVERAL large slots: a grow-only slot that an earlier call left far larger than this call needs (the 230 GB
// pipeline pool of a materialised Forward, then an E-step that wants 2 x 115 GB) would starve the other slot -- ws_get hands the big
// one out as it is, pins it, and the second request finds no memory.  When the growth the listed requests need does not fit beside
// what is allocated, the listed slots that are larger than their request are released first (one re-allocation, not a failure).
void ws_plan(int n, const int *slots, const size_t *bytes) {
  size_t growth = 0, reclaim = 0;
  for (int k = 0; k < n; ++k) {
    const Workspace &w = g_ws[slots[k]];
    if (!w.p || w.bytes < bytes[k]) { growth += std::max<size_t>(bytes[k], 256); reclaim += w.p ? w.bytes : 0; }
  }
  if (!growth) return;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return; }
  if (freeB + reclaim >= growth + ((size_t)256 << 20)) return;
  for (int k = 0; k < n; ++k) {
    Workspace &w = g_ws[slots[k]];
    if (w.p && !w.pinned && w.bytes > bytes[k]) { ws_free_slot(w); ++g_alloc.evictions; }
  }
}

// Small per-call device buffers (pair descriptors, log-likelihoods, offsets): a hipMalloc / hipFree pair per call is
// normally microseconds but now and then tens of milliseconds (measured: 23 ms ahead of a 25 ms Viterbi batch), so freed
// blocks are kept by size class and handed out again.
struct SmallBlock { void *p; size_t bytes; };
static std::vector<SmallBlock> g_smallFree, g_smallLive;
hipError_t sm_alloc(void **out, size_t bytes) {
  size_t cls = 4096;
  while (cls < bytes) cls <<= 1;
  for (size_t k = 0; k < g_smallFree.size(); ++k)
    if (g_smallFree[k].bytes == cls) {
      *out = g_smallFree[k].p;
      g_smallLive.push_back(g_smallFree[k]);
      g_smallFree.erase(g_smallFree.begin() + k);
      return hipSuccess;
    }
  const hipError_t e = hipMalloc(out, cls);
  if (e == hipSuccess) g_smallLive.push_back({*out, cls});
  return e;
}
void sm_free(void *p) {
  if (!p) return;
  for (size_t k = 0; k < g_smallLive.size(); ++k)
    if (g_smallLive[k].p == p) {
      if (g_smallLive[k].bytes <= ((size_t)64 << 20) && g_smallFree.size() < 32) g_smallFree.push_back(g_smallLive[k]);
      else (void)hipFree(p);
      g_smallLive.erase(g_smallLive.begin() + k);
      return;
    }
  (void)hipFree(p);
}

// Large device -> host copies into caller-owned (pageable) memory go through two pinned staging buffers: a hipMemcpy
// straight into fresh pageable pages pins them on the fly, which now and then costs tens of milliseconds for a 15 MB
// path array; the staged copy runs at the PCIe rate and overlaps the host-side memcpy of chunk k with the DMA of k+1.
static void *g_pinned[2] = {nullptr, nullptr};
static const size_t PINNED_CHUNK = (size_t)8 << 20;
static int d2h_large(void *dst, const void *srcDev, size_t bytes) {
  if (bytes < ((size_t)1 << 20)) {
    if (!hip_ok(hipMemcpyAsync(dst, srcDev, bytes, hipMemcpyDeviceToHost, g_stream), "D2H")) return 1;
    return hip_ok(hipStreamSynchronize(g_stream), "D2H") ? 0 : 1;
  }
  for (int k = 0; k < 2; ++k)
    if (!g_pinned[k] && !hip_ok(hipHostMalloc(&g_pinned[k], PINNED_CHUNK, hipHostMallocDefault), "hipHostMalloc(staging)")) return 1;
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; ++k) if (!hip_ok(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming), "hipEventCreate")) return 1;
  const size_t nChunks = (bytes + PINNED_CHUNK - 1) / PINNED_CHUNK;
  int rc = 0;
  auto issue = [&](size_t c) {
    const size_t off = c * PINNED_CHUNK, n = std::min(PINNED_CHUNK, bytes - off);
    if (!hip_ok(hipMemcpyAsync(g_pinned[c & 1], (const char *)srcDev + off, n, hipMemcpyDeviceToHost, g_stream), "D2H (staged)")) return 1;
    return hip_ok(hipEventRecord(ev[c & 1], g_stream), "hipEventRecord") ? 0 : 1;
  };
  rc = issue(0);
  for (size_t c = 0; c < nChunks && !rc; ++c) {
    if (c + 1 < nChunks) rc = issue(c + 1);
    if (rc || !hip_ok(hipEventSynchronize(ev[c & 1]), "hipEventSynchronize")) { rc = 1; break; }
    const size_t off = c * PINNED_CHUNK, n = std::min(PINNED_CHUNK, bytes - off);
    std::memcpy((char *)dst + off, g_pinned[c & 1], n);
  }
  if (rc) (void)hipStreamSynchronize(g_stream);
  for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ev[k]);
  return rc;
}

int h2d_large(void *dstDev, const void *src, size_t bytes) {
  if (bytes < ((size_t)256 << 10)) {
    if (!hip_ok(hipMemcpyAsync(dstDev, src, bytes, hipMemcpyHostToDevice, g_stream), "H2D")) return 1;
    return 0;   // small pageable copies are staged by the runtime itself before the call returns
  }
  for (int k = 0; k < 2; ++k)
    if (!g_pinned[k] && !hip_ok(hipHostMalloc(&g_pinned[k], PINNED_CHUNK, hipHostMallocDefault), "hipHostMalloc(staging)")) return 1;
  for (size_t off = 0, c = 0; off < bytes; off += PINNED_CHUNK, ++c) {
    const size_t n = std::min(PINNED_CHUNK, bytes - off);
    if (c >= 2 && !hip_ok(hipStreamSynchronize(g_stream), "H2D (staged)")) return 1;   // the buffer about to be overwritten has been read
    std::memcpy(g_pinned[c & 1], (const char *)src + off, n);
    if (!hip_ok(hipMemcpyAsync((char *)dstDev + off, g_pinned[c & 1], n, hipMemcpyHostToDevice, g_stream), "H2D (staged)")) return 1;
  }
  return hip_ok(hipStreamSynchronize(g_stream), "H2D (staged)") ? 0 : 1;
}

static void ws_release() {
  for (Workspace &w : g_ws) ws_free_slot(w);
  for (SmallBlock &b : g_smallFree) (void)hipFree(b.p);
  g_smallFree.clear();
  for (int k = 0; k < 2; ++k) if (g_pinned[k]) { (void)hipHostFree(g_pinned[k]); g_pinned[k] = nullptr; }
}

static int env_flag_default(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}

size_t budget_bytes() {
  if (g_mem_budget) return g_mem_budget;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return (size_t)8 << 30;
  // STICKY: what is free + cached moves by megabytes from call to call (a batch's tokens, a machine's record tables, a small block
  // released), chunk sizes and the pipeline's pool follow the budget, and ws_get re-allocates a pool for a request that grew by ANY
  // amount -- a 230 GB hipFree + hipMalloc pair is seconds (round 5: 7.1 s of wall around 75 ms of kernels in every call that
  // alternated with the one-tape sweeps).  So the budget only ever follows the memory DOWN, or up by more than an eighth.
  // MB_MEM_FRACTION (default 0.90): the share of the device this process may fill -- several ranks on ONE device (the gloo dry runs of
  // bench.py / boss.py, or a host that co-locates processes) must not each claim it.  0.90 and not the earlier 0.80: BASELINE config 5's
  // E-step at 64 x 50 kb needs 2 x 129.6 GB of matrices, which 0.80 of a 309 GB device cuts into two chunks of 32 sequences (555 ms;
  // one chunk: 518 ms); what the library allocates outside the pools (tokens, programs, exchange buffers) is well under 1 GB
  static size_t sticky = 0;
  static double stickyFrac = 0.0;
  double frac = 0.90;
  if (const char *e = opt_env("MB_MEM_FRACTION")) { const double f = atof(e); if (f > 0.0 && f <= 0.95) frac = f; }
  const size_t cur = (size_t)((double)(freeB + cached_bytes()) * frac);
  if (!sticky || frac != stickyFrac || cur < sticky || cur > sticky + sticky / 8 || !env_flag_default("MB_POOL_STICKY", 1)) { sticky = cur; stickyFrac = frac; }
  return sticky;
}

// kernel launchers implemented in the kernel files
int launch_generic_fill(const mb_machine *, int, const PairDesc *, long long, const int *, const int *, double *, int, hipStream_t,
                        const int *, const int *);
int launch_fill_neg_inf(double *, long long, hipStream_t);
int launch_gather_loglike(const PairDesc *, long long, const double *, int, int, double *, hipStream_t);
int launch_generic_counts(const mb_machine *, const PairDesc *, long long, long long, const int *, const int *,
                          const double *, const double *, double *, hipStream_t);
int launch_compact_paths(const uint32_t *, const long long *, const long long *, const long long *, uint32_t *, long long, hipStream_t);
int launch_traceback(const mb_machine *, const PairDesc *, long long, const int *, const int *, const double *,
                     const long long *, uint32_t *, long long *, hipStream_t);
size_t traceback_bytes_lds(const mb_machine *, int);
int launch_traceback_bytes(const mb_machine *, const PairDesc *, long long, const int *, const int *, const unsigned char *, int, const double *,
                           const long long *, uint32_t *, long long *, hipStream_t);

// a second stream for work that is independent of what g_stream runs (created on first use; nullptr if that fails)
static hipStream_t second_stream() {
  static hipStream_t s2 = nullptr;
  static bool tried = false;
  if (!tried) { tried = true; if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) s2 = nullptr; }
  return s2;
}

// an error path is about to release what launched kernels may still read: wait for both streams (their own errors are not news here)
static void quiesce_streams() {
  if (hipStream_t s2 = second_stream()) (void)hipStreamSynchronize(s2);
  (void)hipStreamSynchronize(g_stream);
  (void)hipGetLastError();
}

static int device_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
  }
  return cus;
}

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  bool ok = false;
  Timer() { ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
  ~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  void start() { if (ok) (void)hipEventRecord(a, g_stream); }
  double stop() {
    if (!ok) return 0.0;
    (void)hipEventRecord(b, g_stream);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
  }
};

// Split [0,nPairs) into chunks whose matrices (nMatrices per pair, doubles) fit the budget.
struct Chunk { long long p0, p1, cells; };
static bool plan_chunks(const mb_batch *b, int nMatrices, std::vector<Chunk> &out, double bytesPerCell = 0.0) {
  const size_t budget = budget_bytes();
  long long maxCells = bytesPerCell > 0.0 ? (long long)((double)budget / bytesPerCell) : (long long)(budget / (8ull * nMatrices));
  // balance: the same number of chunks as a greedy fill needs, but of even size (a short last chunk would leave
  // most of the chip idle)
  if (b->totalCells > maxCells && maxCells > 0) {
    const long long nChunks = (b->totalCells + maxCells - 1) / maxCells;
    const long long even = (b->totalCells + nChunks - 1) / nChunks + b->maxPairCells;
    if (even < maxCells) maxCells = std::max(even, b->maxPairCells);
  }
  // (a chunk also closes when it has reached its even share of what is LEFT: 64 equal pairs in two chunks are 32 + 32, not 33 + 31 -- the
  //  one-tape E-step runs k = CUs / pairs workgroups per sequence, and 33 pairs get 3 where 32 get 4)
  long long p0 = 0, acc = 0, left = b->totalCells;
  long long chunksLeft = maxCells > 0 ? std::max<long long>(1, (b->totalCells + maxCells - 1) / maxCells) : 1;
  for (long long p = 0; p < b->nPairs; ++p) {
    const PairDesc &pd = b->pairs[p];
    const long long c = (long long)(pd.inLen + 1) * (pd.outLen + 1) * b->m->S;
    if (c > maxCells) {
      set_error("a single DP matrix (" + std::to_string(bytesPerCell > 0.0 ? (long long)((double)c * bytesPerCell) : c * 8ll * nMatrices) + " bytes) exceeds the device memory budget");
      return false;
    }
    const bool shareReached = chunksLeft > 1 && acc > 0 && (acc + c / 2) * chunksLeft > left + acc;      // (left + acc = cells from this chunk's start on)
    if (acc + c > maxCells || shareReached) { out.push_back({p0, p, acc}); p0 = p; acc = 0; chunksLeft = std::max<long long>(1, chunksLeft - 1); }
    acc += c; left -= c;
  }
  if (b->nPairs > p0) out.push_back({p0, b->nPairs, acc});
  return true;
}

// Upload PairDescs of a chunk with cellBase rebased to the chunk's pool.
// tbStride > 0: traceback bytes instead of cells -- cellBase = BYTE offset, tbStride bytes per supercell
static int upload_chunk_descs(const mb_batch *b, const Chunk &c, PairDesc **d_out, std::vector<PairDesc> &tmp, int tbStride = 0) {
  tmp.assign(b->pairs.begin() + c.p0, b->pairs.begin() + c.p1);
  long long base = 0;
  for (auto &pd : tmp) { pd.cellBase = base; base += (long long)(pd.inLen + 1) * (pd.outLen + 1) * (tbStride > 0 ? tbStride : b->m->S); }
  MB_HIP(sm_alloc((void **)d_out, tmp.size() * sizeof(PairDesc)));
  if (!hip_ok(hipMemcpyAsync(*d_out, tmp.data(), tmp.size() * sizeof(PairDesc), hipMemcpyHostToDevice, g_stream), "H2D pair descriptors") ||
      !hip_ok(hipStreamSynchronize(g_stream), "H2D pair descriptors")) { sm_free(*d_out); *d_out = nullptr; return 1; }
  return 0;
}


// ---- kernel-family state attached to a machine ------------------------------------------------------------
struct FastState {
  bool tried = false, mediumOk = false;
  bool exactOk = false;      // one-tape machines: only the exact (levelled) tiled program is built, for Viterbi (onetape_tiled_viterbi)
  bool exactTried = false;
  int G = 0;
  // exact (leveled) programs: Viterbi and Forward with a custom start state; "sum" programs: Forward / Backward,
  // silent closure when it stays small, otherwise the exact program of that direction
  MedProgram fwdExact, fwdSum, bwdSum;
  MedGeom geoFE, geoFS, geoBS;
  // Viterbi with traceback bytes: the exact program itself, or -- when that one splits high-degree states -- an unsplit twin
  bool tbTried = false, tbOk = false;
  MedProgram fwdTb;
  MedGeom geoTb;
  // Forward fill fused with posterior counts (run-time specialised kernel only)
  bool countOk = false;
  MedProgram fwdCnt;
  MedGeom geoCnt;
  // large one-tape machines (mb_wide.hip): programs built on first use, rebuilt after a weight update
  WideProgram wFwd, wBwd, wVit;
  WideProgram wVitTb;        // ... the max program that keeps one traceback code per cell (`--align` without the fp64 matrix)
  bool wVitTbTried = false;
  WideCountPlan wCnt;        // posterior counts of one-tape machines (any size): lane = transition
  UsagePlan usage;           // tiled family: posterior counts as a third pass over two materialised matrices (mb_usage.hip)
  WideTbPlan wTb;            // Viterbi traceback of one-tape machines too large for the LDS edge tables of mb_generic.hip
  // machines with a handful of states (mb_small.cpp): lane = column, states in registers
  bool smallTried = false, smallOk = false;
  SmallProgram smF, smB;
};

static int env_int(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}

// what the closure chooser minimises (fast_state) and what MB_MEDIUM_JIT_VERBOSE prints (mb_debug_jit_source): candidate slots + the
// price of a round and of a synchronisation point in slots (MB_MEDIUM_ROUND_COST, default 0; MB_MEDIUM_SYNC_COST, default 1)
static long long medium_program_cost(const MedProgram &P) {
  const int syncCost = env_int("MB_MEDIUM_SYNC_COST", 1), roundCost = env_int("MB_MEDIUM_ROUND_COST", 0);
  long long c = 0;
  for (const MedRoundInfo &ri : P.roundInfo) c += (long long)ri.slots.size() + roundCost + (ri.sync ? syncCost : 0);
  return c;
}

static FastState *fast_state(mb_machine *m) {
  if (!m->fast) m->fast = new FastState();
  FastState *f = (FastState *)m->fast;
  if (!f->tried) {
    f->tried = true;
    // tiny machines run with 8 columns per wavefront (8 lanes per supercell); 1-state machines stay generic
    if (m->S >= env_int("MB_MEDIUM_MIN_STATES", 2) && m->S <= 4096 && !wide_applicable(m)) {
      int G = env_int("MB_MEDIUM_G", 0);
      if (!medium_valid_G(G)) G = medium_default_G(m->S);
      f->G = G;
      long long nSilent = 0;
      for (long long e = 0; e < m->nTrans; ++e) nSilent += (m->inTok[e] == 0 && m->outTok[e] == 0);
      const bool wantClosure = env_int("MB_MEDIUM_CLOSURE", 1) != 0;
      // The silent closure trades synchronisation points (one wave-level LDS round trip per silent level) for
      // candidates.  Closing ALL levels at once is right for psw2dna (10 levels, 146 silent edges -> 360 pairs), but on
      // machines whose silent sub-graph is deep and dense (protpsw.translate.dnapsw: 22 levels, 551 silent edges ->
      // 12 162 pairs) neither extreme is: the levels are closed in K groups, K chosen by a cost model
      // (candidate slots + 6 per synchronisation point), K = 0 meaning the levelled program.
      // (round 4: a synchronisation point is priced at ONE slot, not six -- since the rounds of a stage issue their loads together
      // (mb_medium_jit.cpp) the kernels are bound by vector issue, i.e. by candidates; measured on protpsw.translate.dnapsw: 12 level
      // groups with 54 slots beat 5 with 75, rolling Forward 398 -> 432 G cells/s, psw2dna's choice does not change)
      auto cost = [](const MedProgram &P) { return medium_program_cost(P); };
      // Beyond even level groups: explicit stage boundaries, grown greedily -- the cut that lowers the cost most is added
      // until none does (protpsw.translate.dnapsw, 22 levels: even K = 3 costs 86 slots, cuts at levels 7 and 14 cost 73;
      // psw2dna Backward: 96 -> 80 with one cut at level 2).  Returns K (0 = levelled) and fills `cuts` when cuts win.
      auto choose = [&](bool backward, std::vector<int> &cuts) {
        cuts.clear();
        int bestK = 0; long long best = -1;
        const int forced = env_int("MB_MEDIUM_CLOSURE_STAGES", -1);
        if (forced >= 0) return forced;
        if (!wantClosure) return 0;
        const int nLev = backward ? m->nLevB : m->nLevF;
        const bool verbose = opt_env("MB_MEDIUM_JIT_VERBOSE") != nullptr;
        auto tryBuild = [&](int K, const std::vector<int> &cs, long long &c) {
          MedProgram P; MedGeom g;
          medium_set_cuts(cs);
          const bool ok = medium_build_host(m, backward, K, G, P, g);
          medium_set_cuts({});
          if (!ok) return false;
          if (K && P.nPairs > 16 * nSilent + 4 * m->S) return false;      // record tables would not stay cache-resident
          c = cost(P);
          return true;
        };
        for (int K = 0; K <= std::min(std::max(nLev - 1, 1), 12); ++K) {
          long long c;
          if (!tryBuild(K, {}, c)) continue;
          if (verbose) fprintf(stderr, "[mbhip] %s program, closure stages %d: cost %lld\n", backward ? "backward" : "forward", K, c);
          if (best < 0 || c < best) { best = c; bestK = K; }
        }
        if (env_int("MB_MEDIUM_CUTS_SEARCH", 1) && nLev > 2 && !opt_env("MB_MEDIUM_CUTS")) {
          std::vector<int> cur;
          long long curCost;
          if (tryBuild(1, {}, curCost)) {
            const int step = std::max(1, (nLev - 2 + 31) / 32);
            for (int it = 0; it < 6; ++it) {
              int bestCut = -1; long long bestC = curCost;
              for (int c = 2; c < nLev; c += step) {
                if (std::find(cur.begin(), cur.end(), c) != cur.end()) continue;
                std::vector<int> t = cur; t.push_back(c);
                long long cc;
                if (tryBuild(1, t, cc) && cc < bestC) { bestC = cc; bestCut = c; }
              }
              if (bestCut < 0) break;
              cur.push_back(bestCut); curCost = bestC;
            }
            std::sort(cur.begin(), cur.end());
            if (verbose) { fprintf(stderr, "[mbhip] %s program, stage cuts at levels", backward ? "backward" : "forward"); for (int c : cur) fprintf(stderr, " %d", c); fprintf(stderr, ": cost %lld (even groups: %lld)\n", curCost, best); }
            if (!cur.empty() && (best < 0 || curCost < best)) { cuts = cur; return (int)cur.size() + 1; }
          }
        }
        return bestK;
      };
      std::vector<int> cutsFwd; int KFwd = 0;   // the Forward program's closure: the count program's fill rounds use the same
      auto buildWith = [&](bool backward, MedProgram &P, MedGeom &geo) {
        std::vector<int> cuts;
        const int K = choose(backward, cuts);
        if (!backward) { cutsFwd = cuts; KFwd = K; }
        medium_set_cuts(cuts);
        const bool okb = medium_build(m, backward, K, G, P, geo);
        medium_set_cuts({});
        return okb;
      };
      bool ok = medium_build(m, false, 0, G, f->fwdExact, f->geoFE);
      if (ok) ok = buildWith(false, f->fwdSum, f->geoFS);
      if (ok) ok = buildWith(true, f->bwdSum, f->geoBS);
      f->mediumOk = ok;
      f->exactOk = ok;
      if (ok && env_int("MB_MEDIUM_COUNTS", 1)) {
        int Gc = env_int("MB_MEDIUM_COUNT_G", 0);
        if (!medium_valid_G(Gc)) Gc = env_int("MB_MEDIUM_G", 0) ? G : medium_default_count_G(m->S);
        f->countOk = medium_build_count(m, Gc, f->fwdCnt, f->geoCnt, KFwd, cutsFwd);
      }
    }
  }
  return f;
}

// The small-machine family: preferred whenever the machine qualifies (mb_set_kernel: 0 auto, 2 forces it; 1 / 3 skip it).
static bool use_small(mb_machine *m) {
  if (g_kernel_choice == 1 || g_kernel_choice == 3) return false;
  if (!small_eligible(m)) return false;
  if (!m->fast) m->fast = new FastState();
  FastState *f = (FastState *)m->fast;
  if (!f->smallTried) {
    f->smallTried = true;
    if (env_int("MB_SMALL", 1)) f->smallOk = small_build(m, false, f->smF) && small_build(m, true, f->smB);
  }
  return f->smallOk;
}

// the program of the traceback-byte Viterbi sweep of the tiled family (nullptr: the machine keeps the fp64 matrix)
static MedProgram *medium_tb_program(mb_machine *m, MedGeom **geo) {
  FastState *f = fast_state(m);
  if (!f->mediumOk) return nullptr;
  if (!f->fwdExact.hasSplits) { *geo = &f->geoFE; return medium_tb_eligible(m, f->fwdExact) ? &f->fwdExact : nullptr; }
  if (!f->tbTried) {
    f->tbTried = true;
    f->tbOk = medium_build_unsplit(m, f->G, f->fwdTb, f->geoTb) && medium_tb_eligible(m, f->fwdTb);
  }
  *geo = &f->geoTb;
  return f->tbOk ? &f->fwdTb : nullptr;
}

static bool use_medium(mb_machine *m) {
  if (g_kernel_choice == 1) return false;
  FastState *f = fast_state(m);
  if (g_kernel_choice == 3 && !f->mediumOk) return false;
  return f->mediumOk;
}

// the one-tape family: program of a direction / semiring, (re)built from the current weights when needed
static WideProgram *wide_program(mb_machine *m, int mode) {
  if (g_kernel_choice == 1 || !wide_applicable(m)) return nullptr;
  FastState *f = fast_state(m);
  WideProgram &P = mode == MB_VITERBI ? f->wVit : (mode == MB_BACKWARD ? f->wBwd : f->wFwd);
  if ((!P.ok || P.dirty) && !wide_build(m, mode == MB_BACKWARD, mode == MB_VITERBI, P)) return nullptr;
  return &P;
}

// One-tape machine of moderate size whose max program has no retimed form (762 states: retimed 44 G cells/s, tiles 32; 1268
// states: 73 vs 22): the log-sum-exp sweeps belong to the one-tape family, but its column-by-column Viterbi sweep walks the
// silent levels one record at a time (0.2 us per level), where the run-time specialised tile kernel has them as straight-line
// code (762 states: 32 vs 14.5 G cells/s, 1268 states: 22.8 vs 15.6).  Decided when a Viterbi fill asks, and again after a weight
// update: which edges are -inf decides whether the retimed form exists (they leave the retiming graph), so a machine can gain or
// lose it.  Callers that never run Viterbi build neither program.
static bool onetape_tiled_viterbi(mb_machine *m) {
  if (!wide_applicable(m) || g_kernel_choice == 1 || m->S > env_int("MB_WIDE_VITERBI_MIN_STATES", 2048)) return false;
  FastState *f = fast_state(m);
  WideProgram *W = wide_program(m, MB_VITERBI);      // (built on first use, rebuilt when the weights changed)
  if (W && W->retOk) return false;
  if (!f->exactTried) {
    f->exactTried = true;
    int G = env_int("MB_MEDIUM_G", 0);
    if (!medium_valid_G(G)) G = medium_default_G(m->S);
    f->G = G;
    f->exactOk = medium_build(m, false, 0, G, f->fwdExact, f->geoFE);
  }
  return f->exactOk;
}

// One-tape machines, `--viterbi / --align` with paths: the retimed max sweep keeping ONE traceback code per cell (mb_wide.hip,
// WideProgram::tbCodes) -- nullptr when the machine has no such program (no retimed form, a fan-in beyond 256, more than 2^15
// states or 2^16 transitions, a silent transition i -> j with j <= i -- the self-loop on state 0 is the one the reference lets
// through: a candidate of its traceback but not of the fill --, MB_ONETAPE_TB=0): the fp64 matrix and its walkers take it then.
// Decided at the first call and AGAIN after every weight update (which edges are -inf decides whether a retimed form exists,
// see onetape_tiled_viterbi): a failed build is retried then, not latched; a program without a retimed form or without codes is
// freed at once (its column-by-column twin would never run).
static WideProgram *wide_tb_program(mb_machine *m) {
  if (g_kernel_choice == 1 || !wide_applicable(m) || !env_int("MB_ONETAPE_TB", 1)) return nullptr;
  FastState *f = fast_state(m);
  WideProgram &P = f->wVitTb;
  if (!f->wVitTbTried || P.dirty) {
    f->wVitTbTried = true;
    P.dirty = false;      // (this set of weights has been looked at, whatever comes of it)
    for (long long e = 0; e < m->nTrans; ++e)
      if (m->inTok[e] == 0 && m->outTok[e] == 0 && m->dst[e] <= m->src[e]) { if (P.ok) wide_free(P); return nullptr; }
    P.tbCodes = true;
    const bool built = wide_build(m, false, true, P);
    if (!built || !(P.ok && P.retOk && P.tbOk)) { wide_free(P); P.dirty = false; return nullptr; }
  }
  return (P.ok && P.retOk && P.tbOk) ? &P : nullptr;
}

// Fill the matrices of one chunk of pairs (materialised), choosing the kernel family.
static int fill_chunk(mb_machine *m, int mode, const PairDesc *d_desc, const std::vector<PairDesc> &hp, const int *d_in,
                      const int *d_out, double *pool, int startState, const mb_batch *b) {
  const bool env = b && b->hasEnv;   // envelopes: cells outside keep the -inf written here (tiles outside them do not even run)
  if (env) {
    long long cells = 0;
    for (const PairDesc &pd : hp) cells = std::max(cells, pd.cellBase + (long long)(pd.inLen + 1) * (pd.outLen + 1) * m->S);
    if (launch_fill_neg_inf(pool, cells, g_stream)) return 1;
  }
  const bool tiledViterbi = !env && mode == MB_VITERBI && onetape_tiled_viterbi(m);
  if (!env && startState == 0 && wide_applicable(m) && g_kernel_choice != 1 && !tiledViterbi) {
    WideProgram *W = wide_program(m, mode);
    if (!W) return 1;
    const int rcW = wide_fill(m, *W, d_desc, (long long)hp.size(), m->nOut ? d_out : d_in, pool, nullptr, g_stream, false, hp.data(), device_cus());
    g_last_kernel = wide_kernel_name(*W);
    return rcW;
  }
  if ((use_medium(m) && !(env && wide_applicable(m))) || tiledViterbi) {
    FastState *f = fast_state(m);
    const bool exactFwd = mode == MB_VITERBI || (mode == MB_FORWARD && startState != 0);
    MedProgram &P = mode == MB_BACKWARD ? f->bwdSum : (exactFwd ? f->fwdExact : f->fwdSum);
    const MedGeom &geo = mode == MB_BACKWARD ? f->geoBS : (exactFwd ? f->geoFE : f->geoFS);
    MedEnv me;
    if (env) { me.d_start = b->d_envStart; me.d_end = b->d_envEnd; me.h_start = b->h_envStart.data(); me.h_end = b->h_envEnd.data(); }
    const int rc = medium_fill_materialised(m, P, geo, mode == MB_VITERBI ? MB_VITERBI : MB_FORWARD,
                                            (mode == MB_FORWARD && startState != 0) ? startState : -1, d_desc, hp, d_in, d_out, pool, g_stream, me);
    if (rc >= 0) {
      g_last_kernel = medium_jit_ready(P, mode == MB_VITERBI ? MB_VITERBI : MB_FORWARD, MED_MAT_FULL) ? "k_medium_jit" : (mode == MB_VITERBI ? "k_medium_tile<1>" : "k_medium_tile<0>");
      return rc;
    }
    // (-1: envelopes need the run-time specialised kernel and it is unavailable: the generic family takes the chunk)
  }
  g_last_kernel = mode == MB_VITERBI ? "k_generic_fill_fwd<1>" : (mode == MB_BACKWARD ? "k_generic_fill_bwd" : "k_generic_fill_fwd<0>");
  return launch_generic_fill(m, mode, d_desc, (long long)hp.size(), d_in, d_out, pool, startState, g_stream,
                             env ? b->d_envStart : nullptr, env ? b->d_envEnd : nullptr);
}

// ---- the small-machine family (mb_small.cpp): placement of its per-pair buffers and chunking -----------------------------
struct SmallPlan {
  std::vector<SmAux> aux;
  long long poolD = 0, tbB = 0, haloD = 0, boundD = 0;
};

// Halo rows and boundary records are laid out for the LARGER of the two sweep directions' needs: a count call runs the
// Backward and the Forward program over the same buffers (and offsets), and the two programs of an asymmetric machine
// keep different numbers of values per halo row / boundary record.
static int small_halo_width(const mb_machine *m) { const FastState *f = (const FastState *)m->fast; return std::max(1, std::max(f->smF.H, f->smB.H)); }
static int small_bound_width(const mb_machine *m) { const FastState *f = (const FastState *)m->fast; return std::max(f->smF.NBD, f->smB.NBD); }

static void small_plan(const mb_machine *m, const SmallProgram &P, const PairDesc *hp, long long n, bool wantPool, bool wantTb, SmallPlan &pl) {
  const int HW = small_halo_width(m), BW = small_bound_width(m);
  pl = SmallPlan();
  pl.aux.resize((size_t)n);
  for (long long k = 0; k < n; ++k) {
    const PairDesc &pd = hp[k];
    SmAux &a = pl.aux[(size_t)k];
    a.pool = pl.poolD; a.tb = pl.tbB; a.halo = pl.haloD; a.bound = pl.boundD;
    if (wantPool) pl.poolD += small_pair_doubles(P.S, pd.inLen, pd.outLen);
    if (wantTb) pl.tbB += (small_pair_tb_bytes(P.S, pd.inLen, pd.outLen) + 15) & ~15ll;
    pl.haloD += (long long)small_strips(pd.inLen) * (pd.outLen + 1) * HW;   // one halo column per strip
    pl.boundD += (long long)small_strips(pd.inLen) * 64 * BW;
  }
}

// Split the batch into chunks whose buffers fit the memory budget (pairs are independent).
static bool small_chunks_plan(const mb_batch *b, const SmallProgram &P, bool wantPool, bool wantTb, std::vector<Chunk> &out) {
  const long long budget = (long long)budget_bytes();
  long long p0 = 0, acc = 0;
  for (long long p = 0; p < b->nPairs; ++p) {
    const PairDesc &pd = b->pairs[p];
    long long c = ((long long)small_strips(pd.inLen) * (pd.outLen + 1) * small_halo_width(b->m) + (long long)small_strips(pd.inLen) * 64 * small_bound_width(b->m)) * 8;
    if (wantPool) c += small_pair_doubles(P.S, pd.inLen, pd.outLen) * 8;
    if (wantTb) c += small_pair_tb_bytes(P.S, pd.inLen, pd.outLen) + 16;
    if (c > budget) { set_error("a single DP matrix (" + std::to_string(c) + " bytes) exceeds the device memory budget"); return false; }
    if (acc + c > budget) { out.push_back({p0, p, acc}); p0 = p; acc = 0; }
    acc += c;
  }
  if (b->nPairs > p0) out.push_back({p0, b->nPairs, acc});
  return true;
}

// buffers of one chunk: aux records uploaded, workspaces sized; fills `sw`
static int small_prepare(mb_batch *b, const Chunk &c, const SmallProgram &P, bool wantPool, bool wantTb, SmallPlan &pl,
                         std::vector<PairDesc> &hp, SmAux **d_aux, SmSweep &sw, size_t chunkNo = (size_t)-1) {
  hp.assign(b->pairs.begin() + c.p0, b->pairs.begin() + c.p1);
  small_plan(b->m, P, hp.data(), (long long)hp.size(), wantPool, wantTb, pl);
  MB_HIP(sm_alloc((void **)d_aux, hp.size() * sizeof(SmAux)));   // (the caller frees *d_aux on every path)
  MB_HIP(hipMemcpyAsync(*d_aux, pl.aux.data(), hp.size() * sizeof(SmAux), hipMemcpyHostToDevice, g_stream));
  sw = SmSweep();
  sw.d_pairs = b->d_pairs + c.p0; sw.pairs = &hp; sw.d_in = b->d_in; sw.d_out = b->d_out; sw.d_aux = *d_aux;
  if (chunkNo != (size_t)-1) {
    if (b->smTiles.size() <= 2 * chunkNo + 1) b->smTiles.resize(2 * chunkNo + 2);
    for (int bw = 0; bw < 2; ++bw) {
      SmTileCache &tc = b->smTiles[2 * chunkNo + bw];
      if (tc.p0 != c.p0 || tc.p1 != c.p1 || tc.envVersion != b->envVersion) {
        if (tc.d_tiles) (void)hipFree(tc.d_tiles);
        if (tc.d_deps) (void)hipFree(tc.d_deps);
        if (tc.d_flags) (void)hipFree(tc.d_flags);
        tc.d_deps = tc.d_flags = nullptr;
        tc = SmTileCache(); tc.p0 = c.p0; tc.p1 = c.p1; tc.envVersion = b->envVersion;
      }
    }
    sw.tileCacheFwd = &b->smTiles[2 * chunkNo]; sw.tileCacheBwd = &b->smTiles[2 * chunkNo + 1];
  }
  if (b->hasEnv) {   // PairDesc::envBase (-1: full) indexes them
    sw.d_envStart = b->d_envStart; sw.d_envEnd = b->d_envEnd;
    sw.h_envStart = b->h_envStart.data(); sw.h_envEnd = b->h_envEnd.data();
  }
  sw.haloDoubles = pl.haloD;      // (pre-filled by the sweep: -inf for tiles that do not run, the sentinel of the persistent strips)
  if (wantPool && !(sw.d_pool = (double *)ws_get(0, (size_t)std::max<long long>(pl.poolD, 1) * 8))) return 1;
  if (wantTb && !(sw.d_tb = (unsigned char *)ws_get(8, (size_t)std::max<long long>(pl.tbB, 16)))) return 1;
  if (!(sw.d_halo = (double *)ws_get(9, (size_t)std::max<long long>(pl.haloD, 1) * 8))) return 1;
  if (!(sw.d_bound = (double *)ws_get(10, (size_t)std::max<long long>(pl.boundD, 1) * 8))) return 1;
  return 0;
}

static int small_forward(mb_batch *b, int flags, double *loglike) {
  FastState *f = (FastState *)b->m->fast;
  const bool mat = !(flags & MB_ROLLING);
  std::vector<Chunk> chunks;
  if (!small_chunks_plan(b, f->smF, mat, false, chunks)) return 1;
  double *d_ll = nullptr;
  MB_HIP(sm_alloc((void **)&d_ll, b->nPairs * sizeof(double)));
  int rc = 0;
  Timer tm;
  size_t chunkNo = 0;
  for (const Chunk &c : chunks) {
    SmallPlan pl; std::vector<PairDesc> hp; SmAux *d_aux = nullptr; SmSweep sw;
    if (!(rc = small_prepare(b, c, f->smF, mat, false, pl, hp, &d_aux, sw, chunkNo++))) {
      sw.d_loglike = d_ll + c.p0;
      tm.start();
      rc = small_sweep(f->smF, SM_SUM, mat, sw, g_stream);
      g_last_ms += tm.stop();
    }
    sm_free(d_aux);
    if (rc) break;
  }
  g_last_kernel = small_kernel_name(f->smF, SM_SUM, mat);
  if (!rc && !hip_ok(hipMemcpy(loglike, d_ll, b->nPairs * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) rc = 1;
  sm_free(d_ll);
  return rc;
}

// ViterbiMatrix(eval, sp).logLike() / path(): one traceback byte per cell, walked on the device
static int small_viterbi(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap) {
  FastState *f = (FastState *)b->m->fast;
  const bool wantPaths = pathEdges != nullptr && pathOff != nullptr;
  std::vector<Chunk> chunks;
  if (!small_chunks_plan(b, f->smF, false, true, chunks)) return 1;
  int rc = 0;
  Timer tm;
  long long written = 0;
  const bool timing = opt_env("MB_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tPrev = now();
  auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[mbhip] viterbi %-28s %7.2f ms\n", what, t - tPrev); tPrev = t; } };
  size_t chunkNo = 0;
  for (const Chunk &c : chunks) {
    const long long np = c.p1 - c.p0;
    SmallPlan pl; std::vector<PairDesc> hp; SmAux *d_aux = nullptr; SmSweep sw;
    double *d_ll = nullptr;
    std::vector<long long> slot(np + 1, 0), len(np, 0);
    do {
      if ((rc = small_prepare(b, c, f->smF, false, true, pl, hp, &d_aux, sw, chunkNo++))) break;
      if (!hip_ok(sm_alloc((void **)&d_ll, np * sizeof(double)), "hipMalloc")) { rc = 1; break; }
      sw.d_loglike = d_ll;
      lap("chunk set-up");
      tm.start();
      if ((rc = small_sweep(f->smF, SM_TB, false, sw, g_stream))) break;
      lap("fill (traceback bytes)");
      long long *d_slot = nullptr, *d_len = nullptr; uint32_t *d_path = nullptr;
      if (wantPaths) {
        for (long long p = 0; p < np; ++p) slot[p + 1] = slot[p] + mb_viterbi_path_bound(b->m, hp[p].inLen, hp[p].outLen);
        if (!(d_slot = (long long *)ws_get(3, (np + 1) * sizeof(long long)))) { rc = 1; break; }
        if (!(d_len = (long long *)ws_get(4, np * sizeof(long long)))) { rc = 1; break; }
        if (!(d_path = (uint32_t *)ws_get(5, std::max<long long>(slot[np], 1) * sizeof(uint32_t)))) { rc = 1; break; }
        if (!hip_ok(hipMemcpyAsync(d_slot, slot.data(), (np + 1) * sizeof(long long), hipMemcpyHostToDevice, g_stream), "H2D")) { rc = 1; break; }
        if ((rc = launch_small_traceback(f->smF, sw.d_pairs, np, b->d_in, b->d_out, sw.d_tb, d_aux, d_ll, d_slot, d_path, d_len, g_stream))) break;
      }
      g_last_ms += tm.stop();
      lap("traceback kernel");
      if (!hip_ok(hipMemcpy(loglike + c.p0, d_ll, np * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) { rc = 1; break; }
      if (wantPaths) {
        if (!hip_ok(hipMemcpy(len.data(), d_len, np * sizeof(long long), hipMemcpyDeviceToHost), "D2H path lengths")) { rc = 1; break; }
        std::vector<long long> off(np, -1);
        long long total = 0;
        for (long long p = 0; p < np && !rc; ++p) {
          const long long n = len[p];
          if (n == -1) continue;   // -inf end cell: no path (src/dpmatrix.defs.h:84)
          if (n < 0) { set_error("Viterbi traceback exceeded its path bound"); rc = 1; break; }
          off[p] = total; total += n;
        }
        if (rc) break;
        if (written + total > pathCap) { set_error("pathCap too small for the Viterbi paths of this batch"); rc = 1; break; }
        long long *d_off = (long long *)ws_get(6, np * sizeof(long long));
        uint32_t *d_packed = (uint32_t *)ws_get(7, std::max<long long>(total, 1) * sizeof(uint32_t));
        if (!d_off || !d_packed) { rc = 1; break; }
        if (!hip_ok(hipMemcpyAsync(d_off, off.data(), np * sizeof(long long), hipMemcpyHostToDevice, g_stream), "H2D")) { rc = 1; break; }
        if ((rc = launch_compact_paths(d_path, d_slot, d_len, d_off, d_packed, np, g_stream))) break;
        if (total && d2h_large(pathEdges + written, d_packed, total * sizeof(uint32_t))) { rc = 1; break; }
        if (!hip_ok(hipStreamSynchronize(g_stream), "path compaction")) { rc = 1; break; }
        lap("pack + D2H paths");
        for (long long p = 0; p < np; ++p) {
          if (len[p] > 0) written += len[p];
          pathOff[c.p0 + p + 1] = written;
        }
      }
    } while (0);
    sm_free(d_aux); sm_free(d_ll);
    if (rc) break;
  }
  g_last_kernel = "k_small_tb";
  return rc;
}

// MachineCounts(eval, seqPairList): Backward matrices written once (tile-major), then a Forward sweep that reads them and
// keeps the posterior usage sums on chip -- the Forward matrix itself never reaches HBM.
static const int SMALL_COUNT_REPLICAS = 64;
static int small_counts(mb_batch *b, double *counts, double *loglikeSum, double *loglike) {
  FastState *f = (FastState *)b->m->fast;
  const long long nT = b->m->nTrans;
  std::vector<Chunk> chunks;
  if (!small_chunks_plan(b, f->smB, true, false, chunks)) return 1;
  double *d_rep = nullptr, *d_ll = nullptr, *d_bll = nullptr;
  int rc = 0;
  Timer tm;
  do {
    if (!hip_ok(sm_alloc((void **)&d_rep, std::max<long long>(nT, 1) * SMALL_COUNT_REPLICAS * sizeof(double)), "hipMalloc")) { rc = 1; break; }
    if (!hip_ok(sm_alloc((void **)&d_ll, b->nPairs * sizeof(double)), "hipMalloc")) { rc = 1; break; }
    if (!hip_ok(sm_alloc((void **)&d_bll, b->nPairs * sizeof(double)), "hipMalloc")) { rc = 1; break; }
    if (!hip_ok(hipMemsetAsync(d_rep, 0, std::max<long long>(nT, 1) * SMALL_COUNT_REPLICAS * sizeof(double), g_stream), "memset")) { rc = 1; break; }
    size_t chunkNo = 0;
  for (const Chunk &c : chunks) {
      SmallPlan pl; std::vector<PairDesc> hp; SmAux *d_aux = nullptr; SmSweep sw;
      if (!(rc = small_prepare(b, c, f->smB, true, false, pl, hp, &d_aux, sw, chunkNo++))) {
        tm.start();
        sw.d_loglike = d_bll + c.p0;
        rc = small_sweep(f->smB, SM_SUM, true, sw, g_stream);                 // BackwardMatrix::fill, src/backward.cpp:18-46
        if (!rc) {
          sw.d_loglike = d_ll + c.p0; sw.d_bwdLL = d_bll + c.p0; sw.d_counts = d_rep; sw.nRep = SMALL_COUNT_REPLICAS;
          rc = small_sweep(f->smF, SM_COUNT, false, sw, g_stream);            // Forward + getCounts, src/backward.cpp:58-87
        }
        g_last_ms += tm.stop();
      }
      sm_free(d_aux);
      if (rc) break;
    }
    if (rc) break;
    std::vector<double> hc((size_t)nT * SMALL_COUNT_REPLICAS), hll(b->nPairs);
    if (nT && !hip_ok(hipMemcpy(hc.data(), d_rep, hc.size() * sizeof(double), hipMemcpyDeviceToHost), "D2H counts")) { rc = 1; break; }
    if (!hip_ok(hipMemcpy(hll.data(), d_ll, b->nPairs * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) { rc = 1; break; }
    for (long long e = 0; e < nT; ++e) {
      if (g_deterministic) {      // fixed point, 2^-36: the replicas add up as integers
        unsigned long long tot = 0; bool inRange = true;
        for (int r = 0; r < SMALL_COUNT_REPLICAS; ++r) { unsigned long long u; std::memcpy(&u, &hc[(size_t)r * nT + e], 8); inRange = inRange && u < (1ull << 62); tot += u; }
        double c;
        if (!det_to_double(tot, c) || !inRange) { set_error("MB_DETERMINISTIC: a posterior count left the fixed-point range (6.7e7 per transition and call): split the batch or use the floating-point mode"); rc = 1; break; }
        counts[e] += c;
        continue;
      }
      double s = 0.0;
      for (int r = 0; r < SMALL_COUNT_REPLICAS; ++r) s += hc[(size_t)r * nT + e];
      counts[e] += s;
    }
    double s = 0;
    for (long long p = 0; p < b->nPairs; ++p) { s += hll[p]; if (loglike) loglike[p] = hll[p]; }   // loglike += forward.logLike(), src/counts.cpp:61-62
    if (loglikeSum) *loglikeSum += s;
  } while (0);
  sm_free(d_rep); sm_free(d_ll); sm_free(d_bll);
  g_last_kernel = "k_small_count";
  return rc;
}

// one full matrix in the reference's layout (mb_fill): sweep tile-major, convert on the device, copy out
static int small_fill(mb_batch *b, int mode, double *cellsOut) {
  FastState *f = (FastState *)b->m->fast;
  SmallProgram &P = mode == MB_BACKWARD ? f->smB : f->smF;
  const PairDesc &pd = b->pairs[0];
  const long long n = (long long)(pd.inLen + 1) * (pd.outLen + 1) * b->m->S;
  const Chunk c{0, 1, 0};
  SmallPlan pl; std::vector<PairDesc> hp; SmAux *d_aux = nullptr; SmSweep sw;
  double *d_ll = nullptr, *d_cells = nullptr;
  int rc = 0;
  do {
    if ((size_t)(small_pair_doubles(P.S, pd.inLen, pd.outLen) + n) * 8 > budget_bytes()) { set_error("matrix exceeds the device memory budget"); rc = 1; break; }
    if ((rc = small_prepare(b, c, P, true, false, pl, hp, &d_aux, sw))) break;
    if (!hip_ok(sm_alloc((void **)&d_ll, sizeof(double)), "hipMalloc")) { rc = 1; break; }
    if (!(d_cells = (double *)ws_get(1, (size_t)std::max<long long>(n, 1) * 8))) { rc = 1; break; }
    sw.d_loglike = d_ll;
    if ((rc = small_sweep(P, mode == MB_VITERBI ? SM_MAX : SM_SUM, true, sw, g_stream))) break;
    if ((rc = launch_fill_neg_inf(d_cells, n, g_stream))) break;
    if ((rc = launch_small_unpack(sw.d_pool, P.S, pd.inLen, pd.outLen, mode == MB_BACKWARD, d_cells,
                                  b->hasEnv && pd.envBase >= 0 ? b->d_envStart + pd.envBase : nullptr,
                                  b->hasEnv && pd.envBase >= 0 ? b->d_envEnd + pd.envBase : nullptr, g_stream))) break;
    if (!hip_ok(hipStreamSynchronize(g_stream), "fill kernel")) { rc = 1; break; }
    if (!hip_ok(hipMemcpy(cellsOut, d_cells, n * sizeof(double), hipMemcpyDeviceToHost), "D2H matrix")) { rc = 1; break; }
  } while (0);
  sm_free(d_aux); sm_free(d_ll);
  g_last_kernel = small_kernel_name(P, mode == MB_VITERBI ? SM_MAX : SM_SUM, true);
  return rc;
}

}  // namespace mb

using namespace mb;

extern "C" {

int mb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mb_set_device(int device) {
  ApiGuard guard;
  if (g_init && device != g_device) {   // the stream, the workspaces and every compiled program are bound to the first device
    set_error("mb_set_device: the library is already initialised on device " + std::to_string(g_device) + " (one process drives one GPU)");
    return 1;
  }
  MB_HIP(hipSetDevice(device));
  return 0;
}

int mb_synchronize(void) {
  ApiGuard guard;
  if (!g_init) return 0;      // nothing was ever queued
  MB_HIP(hipDeviceSynchronize());
  return 0;
}

const char *mb_last_error(void) { return g_err.c_str(); }
double mb_last_device_ms(void) { return g_last_ms; }
const char *mb_last_kernel_name(void) { return g_last_kernel; }
int64_t mb_last_launch_count(void) { return g_last_launches; }

int mb_set_kernel(int which) {
  if (which < 0 || which > 3) { set_error("mb_set_kernel: unknown kernel family"); return 1; }
  g_kernel_choice = which;
  return 0;
}

int mb_set_memory_budget(size_t bytes) { g_mem_budget = bytes; return 0; }

int mb_release_workspace(void) {
  ApiGuard guard;
  ws_release();
  return 0;
}

int mb_alloc_stats(int64_t *poolAllocs, int64_t *poolFrees, int64_t *evictions, uint64_t *bytesAllocated, double *ms) {
  ApiLock lock;
  if (poolAllocs) *poolAllocs = g_alloc.allocs;
  if (poolFrees) *poolFrees = g_alloc.frees;
  if (evictions) *evictions = g_alloc.evictions;
  if (bytesAllocated) *bytesAllocated = g_alloc.bytes;
  if (ms) *ms = g_alloc.ms;
  return 0;
}

int mb_jit_stats(double *compileMs, int64_t *compiles, int64_t *cacheHits) {
  if (compileMs) *compileMs = jit_compile_ms();
  if (compiles) *compiles = jit_compiles();
  if (cacheHits) *cacheHits = jit_cache_hits();
  return 0;
}

int mb_machine_sweep_ops(mb_machine *m, double *expPerCell, double *logPerCell, const char **family) {
  ApiGuard guard;
  if (!m) { set_error("null argument"); return 1; }
  double ne = 0.0, nl = 0.0; const char *fam = "generic";
  if (use_small(m)) {
    // per supercell and lane: two candidates -> exp + log of the pair; three or more -> one exp each + one log
    const SmallProgram &P = ((FastState *)m->fast)->smF;
    for (int d = 0; d < P.S; ++d) { const size_t n = P.cand[d].size(); if (n == 2) { ne += 1; nl += 1; } else if (n > 2) { ne += (double)n; nl += 1; } }
    ne /= m->S; nl /= m->S; fam = "small";
  } else if (wide_applicable(m) && g_kernel_choice != 1) {
    WideProgram *W = wide_program(m, MB_FORWARD);
    if (!W) return 1;
    ne = (double)W->slotsPerColumn * W->W / m->S; nl = (double)W->rounds.size() * W->W / m->S; fam = "one-tape";   // one exp per slot and lane (+ the group reduction), one log per round
  } else if (use_medium(m)) {
    const MedProgram &P = fast_state(m)->fwdSum;
    double slots = 0.0, logs = 0.0;   // a round with one candidate slot is a plain add: no exp, no log
    for (const MedRoundInfo &ri : P.roundInfo) if (ri.slots.size() > 1) { slots += (double)ri.slots.size(); logs += 1.0; }
    ne = slots * P.LPG / m->S; nl = logs * P.LPG / m->S; fam = "tiled";
  } else {
    ne = (double)m->nTrans / m->S; nl = ne;   // generic family: log1p(exp()) per candidate, token-filtered at run time (upper bound)
  }
  if (expPerCell) *expPerCell = ne;
  if (logPerCell) *logPerCell = nl;
  if (family) *family = fam;
  return 0;
}

// Tuning knobs are read from the environment when a machine's programs / kernels are built (DESIGN.md section 4.4 lists
// them); this is the same switchboard for a host that prefers calls to environment variables.
int mb_set_option(const char *name, const char *value) {
  ApiLock lock;
  if (!name || strncmp(name, "MB_", 3) != 0) { set_error("mb_set_option: option names start with MB_"); return 1; }
  opt_set(name, value);      // (the library's table: the process environment is the host's, not ours to write)
  return 0;
}

const char *mb_get_option(const char *name) {
  ApiLock lock;
  if (!name || strncmp(name, "MB_", 3) != 0) return nullptr;
  if (strcmp(name, "MB_INFO_INPLACE_RING_KERNELS") == 0) {      // (read-only: kernel kinds that took the in-place ring so far, for tests)
    static thread_local std::string v;
    v = std::to_string(medium_inplace_kernels());
    return v.c_str();
  }
  return opt_env(name);
}

mb_machine *mb_machine_create(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src,
                              const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok, const double *logWeight) {
  ApiGuard guard;
  if (nStates <= 0) { set_error("EvaluatedMachine has no states"); return nullptr; }
  if (nInTok < 0 || nOutTok < 0 || nTrans < 0) { set_error("mb_machine_create: negative size"); return nullptr; }
  if (ensure_init()) return nullptr;
  mb_machine *m = new mb_machine();
  m->S = nStates; m->nIn = nInTok; m->nOut = nOutTok; m->nTrans = nTrans;
  m->src.assign(src, src + nTrans); m->dst.assign(dst, dst + nTrans);
  m->inTok.assign(inTok, inTok + nTrans); m->outTok.assign(outTok, outTok + nTrans);
  m->logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(m, &err)) { set_error(err); delete m; return nullptr; }
  if (!upload_machine(m)) { free_machine_device(m); delete m; return nullptr; }
  return m;
}

int mb_machine_set_weights(mb_machine *m, const double *logWeight) {
  ApiGuard guard;
  if (!m) { set_error("null machine"); return 1; }
  m->logW.assign(logWeight, logWeight + m->nTrans);
  if (!upload_weights(m)) return 1;
  if (m->fast) {
    FastState *f = (FastState *)m->fast;
    if (f->mediumOk && !(medium_refresh_weights(m, f->fwdExact) && medium_refresh_weights(m, f->fwdSum) && medium_refresh_weights(m, f->bwdSum))) return 1;
    if (!f->mediumOk && f->exactOk && !medium_refresh_weights(m, f->fwdExact)) return 1;
    if (f->countOk && !medium_refresh_weights(m, f->fwdCnt)) return 1;
    if (f->tbOk && !medium_refresh_weights(m, f->fwdTb)) return 1;
    if (f->smallOk && !(small_refresh_weights(m, f->smF) && small_refresh_weights(m, f->smB))) return 1;
    f->wFwd.dirty = f->wBwd.dirty = f->wVit.dirty = f->wVitTb.dirty = true;
    usage_free(f->usage);      // (its records carry the weights, and which transitions are -inf: rebuilt by the next count call)
  }
  return 0;
}

void mb_machine_destroy(mb_machine *m) {
  ApiGuard guard;
  if (!m) return;
  if (m->fast) {
    FastState *f = (FastState *)m->fast;
    medium_free(f->fwdExact); medium_free(f->fwdSum); medium_free(f->bwdSum); medium_free(f->fwdCnt); medium_free(f->fwdTb);
    wide_free(f->wFwd); wide_free(f->wBwd); wide_free(f->wVit); wide_free(f->wVitTb); wide_counts_free(f->wCnt); wide_traceback_free(f->wTb);
    small_free(f->smF); small_free(f->smB);
    usage_free(f->usage);
    delete f;
  }
  free_machine_device(m);
  delete m;
}

int32_t mb_machine_n_states(const mb_machine *m) { return m ? m->S : 0; }
int64_t mb_machine_n_trans(const mb_machine *m) { return m ? m->nTrans : 0; }
int32_t mb_machine_n_levels(const mb_machine *m) { return m ? m->nLevF : 0; }

int mb_machine_edge_order(const mb_machine *m, int which, uint32_t *out) {
  if (!m) { set_error("null machine"); return 1; }
  const std::vector<uint32_t> &p = which ? m->outPerm : m->inPerm;
  std::memcpy(out, p.data(), p.size() * sizeof(uint32_t));
  return 0;
}

mb_batch *mb_batch_create(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff, const int32_t *outTok,
                          const int64_t *outOff) {
  ApiGuard guard;
  if (!m) { set_error("null machine"); return nullptr; }
  if (nPairs < 0) { set_error("negative pair count"); return nullptr; }
  mb_batch *b = new mb_batch();
  b->m = m; b->nPairs = nPairs;
  b->pairs.resize(nPairs);
  long long base = 0;
  for (long long p = 0; p < nPairs; ++p) {
    PairDesc &pd = b->pairs[p];
    const long long il = inOff[p + 1] - inOff[p], ol = outOff[p + 1] - outOff[p];
    if (il < 0 || ol < 0 || il > 0x3fffffff || ol > 0x3fffffff) { set_error("bad sequence offsets"); delete b; return nullptr; }
    pd.inBase = inOff[p] - inOff[0]; pd.outBase = outOff[p] - outOff[0];
    pd.inLen = (int)il; pd.outLen = (int)ol; pd.cellBase = base; pd.envBase = -1;
    const long long c = (il + 1) * (ol + 1) * m->S;
    base += c;
    b->maxPairCells = std::max(b->maxPairCells, c);
  }
  b->totalCells = base;
  b->nInTokTotal = nPairs ? inOff[nPairs] - inOff[0] : 0;
  b->nOutTokTotal = nPairs ? outOff[nPairs] - outOff[0] : 0;
  const int32_t *in0 = inTok + (nPairs ? inOff[0] : 0), *out0 = outTok + (nPairs ? outOff[0] : 0);
  // Tokenizer::tokenize throws on symbols outside the alphabet (src/eval.h:33-37); token 0 (epsilon) is not a symbol
  for (long long k = 0; k < b->nInTokTotal; ++k)
    if (in0[k] < 1 || in0[k] > m->nIn) { set_error("Can't tokenize symbol: input token outside the machine's alphabet"); delete b; return nullptr; }
  for (long long k = 0; k < b->nOutTokTotal; ++k)
    if (out0[k] < 1 || out0[k] > m->nOut) { set_error("Can't tokenize symbol: output token outside the machine's alphabet"); delete b; return nullptr; }
  auto fail = [&]() { mb_batch_destroy(b); return (mb_batch *)nullptr; };
  if (!hip_ok(hipMalloc((void **)&b->d_in, std::max<long long>(b->nInTokTotal, 1) * sizeof(int)), "hipMalloc(tokens)")) return fail();
  if (!hip_ok(hipMalloc((void **)&b->d_out, std::max<long long>(b->nOutTokTotal, 1) * sizeof(int)), "hipMalloc(tokens)")) return fail();
  if (!hip_ok(hipMalloc((void **)&b->d_pairs, std::max<long long>(nPairs, 1) * sizeof(PairDesc)), "hipMalloc(pairs)")) return fail();
  if (b->nInTokTotal && !hip_ok(hipMemcpy(b->d_in, in0, b->nInTokTotal * sizeof(int), hipMemcpyHostToDevice), "H2D tokens")) return fail();
  if (b->nOutTokTotal && !hip_ok(hipMemcpy(b->d_out, out0, b->nOutTokTotal * sizeof(int), hipMemcpyHostToDevice), "H2D tokens")) return fail();
  if (nPairs && !hip_ok(hipMemcpy(b->d_pairs, b->pairs.data(), nPairs * sizeof(PairDesc), hipMemcpyHostToDevice), "H2D pairs")) return fail();
  return b;
}

void mb_batch_destroy(mb_batch *b) {
  ApiGuard guard;
  if (!b) return;
  if (b->d_in) (void)hipFree(b->d_in);
  if (b->d_out) (void)hipFree(b->d_out);
  if (b->d_pairs) (void)hipFree(b->d_pairs);
  if (b->d_envStart) (void)hipFree(b->d_envStart);
  if (b->d_envEnd) (void)hipFree(b->d_envEnd);
  for (mb::SmTileCache &tc : b->smTiles) { if (tc.d_tiles) (void)hipFree(tc.d_tiles); if (tc.d_deps) (void)hipFree(tc.d_deps); if (tc.d_flags) (void)hipFree(tc.d_flags); }
  delete b;
}

int64_t mb_batch_cells(const mb_batch *b) { return b ? b->totalCells : 0; }

// Envelope::fits / connected as DPMatrix::alloc asserts them (src/dpmatrix.defs.h:31-32, src/seqpair.cpp:184-193)
static bool env_overlapping(long long s1, long long e1, long long s2, long long e2) { return !(s1 >= e2 || s2 >= e1); }

int mb_batch_set_envelopes(mb_batch *b, const int64_t *envOff, const int32_t *inStart, const int32_t *inEnd) {
  ApiGuard guard;
  if (!b || !envOff) { set_error("null argument"); return 1; }
  if (b->d_envStart) { (void)hipFree(b->d_envStart); b->d_envStart = nullptr; }
  if (b->d_envEnd) { (void)hipFree(b->d_envEnd); b->d_envEnd = nullptr; }
  b->hasEnv = false;
  ++b->envVersion;
  const long long total = envOff[b->nPairs] - envOff[0];
  auto fail = [&](const char *msg) {   // a rejected call leaves the batch with full envelopes, not with half of the new ones
    for (PairDesc &pd : b->pairs) pd.envBase = -1;
    b->hasEnv = false;
    if (b->nPairs) (void)hipMemcpy(b->d_pairs, b->pairs.data(), b->nPairs * sizeof(PairDesc), hipMemcpyHostToDevice);
    set_error(msg);
    return 1;
  };
  for (PairDesc &pd : b->pairs) pd.envBase = -1;
  for (long long p = 0; p < b->nPairs; ++p) {
    PairDesc &pd = b->pairs[p];
    const long long rows = envOff[p + 1] - envOff[p];
    if (rows == 0) continue;   // full envelope
    if (!inStart || !inEnd) return fail("null argument");
    if (rows != (long long)pd.outLen + 1) return fail("Envelope/sequence mismatch");
    const int32_t *st = inStart + envOff[p], *en = inEnd + envOff[p];
    for (long long y = 0; y < rows; ++y)
      if (st[y] < 0 || en[y] > pd.inLen + 1 || st[y] > en[y]) return fail("Envelope/sequence mismatch");
    bool conn = env_overlapping(st[0], en[0], 0, 1);
    for (long long y = 1; conn && y < rows; ++y) conn = env_overlapping(st[y - 1], (long long)en[y - 1] + 1, st[y], en[y]);
    conn = conn && env_overlapping(st[rows - 1], en[rows - 1], pd.inLen, (long long)pd.inLen + 1);
    if (!conn) return fail("Envelope is not connected");
    pd.envBase = envOff[p] - envOff[0];
    b->hasEnv = true;
  }
  if (b->hasEnv) {
    MB_HIP(hipMalloc((void **)&b->d_envStart, std::max<long long>(total, 1) * sizeof(int)));
    MB_HIP(hipMalloc((void **)&b->d_envEnd, std::max<long long>(total, 1) * sizeof(int)));
    MB_HIP(hipMemcpy(b->d_envStart, inStart + envOff[0], total * sizeof(int), hipMemcpyHostToDevice));
    MB_HIP(hipMemcpy(b->d_envEnd, inEnd + envOff[0], total * sizeof(int), hipMemcpyHostToDevice));
    b->h_envStart.assign(inStart + envOff[0], inStart + envOff[0] + total);
    b->h_envEnd.assign(inEnd + envOff[0], inEnd + envOff[0] + total);
  }
  if (b->nPairs) MB_HIP(hipMemcpy(b->d_pairs, b->pairs.data(), b->nPairs * sizeof(PairDesc), hipMemcpyHostToDevice));
  return 0;
}

// ---- Forward ------------------------------------------------------------------------------------------------
static bool rolltiles_ok = true;      // (cleared while run_fill_loglike re-enters itself after the matrix-free tile kernel proved unavailable)
static int run_fill_loglike(mb_batch *b, int mode, int flags, double *loglike) {
  g_last_ms = 0.0; g_last_launches = 0;
  g_last_kernel = "";
  if (b->nPairs == 0) return 0;
  mb_machine *m = b->m;
  if (mode == MB_FORWARD && use_small(m) && small_can_run(((FastState *)m->fast)->smF, SM_SUM, !(flags & MB_ROLLING), b->hasEnv)) return small_forward(b, flags, loglike);
  double *d_ll = nullptr;
  MB_HIP(sm_alloc((void **)&d_ll, b->nPairs * sizeof(double)));
  int rc = 0;
  Timer tm;
  // The rolling sweep runs ONE workgroup per pair: with fewer pairs than CUs the tile pipeline (which cuts every pair
  // into hundreds of tiles and recycles matrix slots) fills the chip better, and only the log-likelihoods are kept
  // either way (psw2dna, 64 pairs: 229 vs 517 G cells/s).
  const bool fewPairs = b->nPairs < env_int("MB_ROLLING_MIN_PAIRS", 192) &&
                        (size_t)b->maxPairCells * 8 * 2 <= budget_bytes();
  if (mode == MB_FORWARD && !b->hasEnv && wide_applicable(m) && g_kernel_choice != 1 &&
      ((flags & MB_ROLLING) || (size_t)b->totalCells * 8 > budget_bytes())) {
    // one-tape machine, log-likelihood only: the two live columns of every sequence stay in LDS, nothing goes to HBM
    WideProgram *W = wide_program(m, MB_FORWARD);
    // Fewer sequences than half the CUs (one workgroup per sequence): every sequence is CUT IN TWO -- Forward over the prefix
    // and Backward over the suffix behind the cut run side by side on two streams, k_onetape_join sums over the emitting
    // transitions that cross the cut.  Same likelihood (a different summation order: ~1e-12 relative), half the sweep length.
    int minLen = 1 << 30;
    for (const PairDesc &pd : b->pairs) minLen = std::min(minLen, m->nOut ? pd.outLen : pd.inLen);
    const bool split = W && env_int("MB_ONETAPE_SPLIT", 1) && 2 * b->nPairs <= device_cus() && minLen >= env_int("MB_ONETAPE_SPLIT_MIN_LEN", 64) && second_stream();
    if (!W) rc = 1;
    else if (split) {
      WideProgram *WB = wide_program(m, MB_BACKWARD);
      const long long n = b->nPairs, S = m->S;
      std::vector<PairDesc> pre, suf;
      auto cutAt = [&](double frac) {
        pre = b->pairs; suf = b->pairs;
        for (long long p = 0; p < n; ++p) {
          const int L = m->nOut ? b->pairs[p].outLen : b->pairs[p].inLen, mid = std::min(L - 1, std::max(0, (int)(L * frac)));
          pre[p].cellBase = suf[p].cellBase = p * S;
          if (m->nOut) { pre[p].outLen = mid; suf[p].outBase += mid + 1; suf[p].outLen = L - mid - 1; }
          else { pre[p].inLen = mid; suf[p].inBase += mid + 1; suf[p].inLen = L - mid - 1; }
        }
      };
      cutAt(0.5);
      // k workgroups per sequence: the two programs' parts are not equally fast (the Backward program of the 5 063-state machine takes 136 ms
      // where the Forward one takes 120): the cut goes where both halves end together under the planner's model
      if (WB) {
        const double cF = wide_parts_cost(m, *W, n, device_cus() / 2, pre.data()), cB = wide_parts_cost(m, *WB, n, device_cus() / 2, suf.data());
        if (cF > 0.0 && cB > 0.0 && env_int("MB_ONETAPE_SPLIT_BALANCE", 1)) cutAt(std::min(0.65, std::max(0.35, cB / (cF + cB))));
      }
      PairDesc *d_pre = nullptr, *d_suf = nullptr;
      double *vec = nullptr;
      hipEvent_t evStart = nullptr, evDone = nullptr;
      do {
        if (!WB) { rc = 1; break; }
        if (!hip_ok(sm_alloc((void **)&d_pre, n * sizeof(PairDesc)), "hipMalloc") || !hip_ok(sm_alloc((void **)&d_suf, n * sizeof(PairDesc)), "hipMalloc")) { rc = 1; break; }
        if (!(vec = (double *)ws_get(2, (size_t)(2 * n * S) * sizeof(double)))) { rc = 1; break; }
        if (!hip_ok(hipMemcpyAsync(d_pre, pre.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, g_stream), "H2D") ||
            !hip_ok(hipMemcpyAsync(d_suf, suf.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, g_stream), "H2D") ||
            !hip_ok(hipStreamSynchronize(g_stream), "H2D")) { rc = 1; break; }   // (pre / suf are pageable host vectors)
        const int *tape = m->nOut ? b->d_out : b->d_in;
        hipStream_t s2 = second_stream();
        tm.start();
        if (hipEventCreateWithFlags(&evStart, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&evDone, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(evStart, g_stream) != hipSuccess || hipStreamWaitEvent(s2, evStart, 0) != hipSuccess) { set_error("one-tape split: stream set-up failed"); rc = 1; break; }
        // (with CUs to spare both halves run k workgroups per sequence, each launch on a stream of its own)
        // (a machine whose ring lives in L2 and whose parts fit the LDS only when a sweep has the whole chip: one half after the other)
        bool parts = wide_parts_for(m, *W, n, device_cus() / 2, pre.data()) > 1 && wide_parts_for(m, *WB, n, device_cus() / 2, suf.data()) > 1;
        const bool oneByOne = !parts && (W->retGv || WB->retGv) && wide_parts_for(m, *W, n, device_cus(), pre.data()) > 1 && wide_parts_for(m, *WB, n, device_cus(), suf.data()) > 1;
        parts = parts || oneByOne;
        const int share = oneByOne ? 1 : 2;
        const int fusedFill = parts ? -1 : wide_fill2(m, *W, *WB, d_pre, d_suf, n, n, tape, vec, vec + n * S, g_stream, true);   // both sweeps in ONE launch
        if (fusedFill > 0) { rc = 1; break; }
        if (fusedFill < 0) {                        // (programs of different kernel variants: two launches on two streams)
          if ((rc = wide_fill(m, *W, d_pre, n, tape, vec, nullptr, oneByOne ? g_stream : s2, true, parts ? pre.data() : nullptr, device_cus() / share))) break;
          if ((rc = wide_fill(m, *WB, d_suf, n, tape, vec + n * S, nullptr, g_stream, true, parts ? suf.data() : nullptr, device_cus() / share))) break;
          if (hipEventRecord(evDone, s2) != hipSuccess || hipStreamWaitEvent(g_stream, evDone, 0) != hipSuccess) { set_error("one-tape split: stream synchronisation failed"); rc = 1; break; }
        }
        if ((rc = wide_join(m, d_pre, n, tape, vec, vec + n * S, d_ll, g_stream))) break;
        { static thread_local std::string nm; nm = std::string(wide_kernel_name(*W)) + " x2 + k_onetape_join"; g_last_kernel = nm.c_str(); }
        g_last_ms += tm.stop();
        if (!hip_ok(hipStreamSynchronize(g_stream), "one-tape forward kernels")) rc = 1;
        if (!rc && wide_parts_failed()) rc = 1;
      } while (0);
      if (rc) { (void)hipStreamSynchronize(g_stream); (void)hipStreamSynchronize(second_stream()); }
      if (evStart) (void)hipEventDestroy(evStart);
      if (evDone) (void)hipEventDestroy(evDone);
      sm_free(d_pre); sm_free(d_suf);
    } else {
      tm.start();
      rc = wide_fill(m, *W, b->d_pairs, b->nPairs, m->nOut ? b->d_out : b->d_in, nullptr, d_ll, g_stream, false, b->pairs.data(), device_cus());
      g_last_kernel = wide_kernel_name(*W);
      g_last_ms += tm.stop();
      if (!rc && !hip_ok(hipStreamSynchronize(g_stream), "one-tape forward kernel")) rc = 1;
      if (!rc && wide_parts_failed()) rc = 1;
    }
  } else if (mode == MB_FORWARD && (flags & MB_ROLLING) && use_medium(m) && !wide_applicable(m) && env_int("MB_MEDIUM_ROLLTILES", 1) &&
             (b->nPairs < env_int("MB_ROLLING_MIN_PAIRS", 192) || b->hasEnv) && rolltiles_ok) {
    // log-likelihoods only, few pairs or envelopes: the tile pipeline WITHOUT a matrix (halo columns + boundary records)
    FastState *f = fast_state(m);
    MedEnv me;
    if (b->hasEnv) { me.d_start = b->d_envStart; me.d_end = b->d_envEnd; me.h_start = b->h_envStart.data(); me.h_end = b->h_envEnd.data(); }
    std::vector<PairDesc> hp(b->pairs);
    rc = launch_fill_neg_inf(d_ll, b->nPairs, g_stream);
    tm.start();
    const int r2 = rc ? rc : medium_forward_rolltiles(m, f->fwdSum, f->geoFS, b->d_pairs, hp, b->d_in, b->d_out, d_ll, g_stream, me);
    g_last_ms += tm.stop();
    if (r2 < 0) { sm_free(d_ll); rolltiles_ok = false; const int r3 = run_fill_loglike(b, mode, flags, loglike); rolltiles_ok = true; return r3; }   // kernel unavailable: the other paths
    rc = r2;
    g_last_kernel = "k_medium_jit";
    if (!rc && !hip_ok(hipStreamSynchronize(g_stream), "rolling forward kernels")) rc = 1;
  } else if (mode == MB_FORWARD && (flags & MB_ROLLING) && !fewPairs && !b->hasEnv && use_medium(m)) {
    // RollingOutputForwardMatrix: no matrix in HBM, only two halo columns per pair
    FastState *f = fast_state(m);
    std::vector<long long> hb(b->nPairs);
    long long tot = 0;
    for (long long p = 0; p < b->nPairs; ++p) { hb[p] = tot; tot += 2ll * (b->pairs[p].outLen + 1) * m->S; }
    double *d_halo = nullptr; long long *d_hb = nullptr;
    do {
      if (!(d_halo = (double *)ws_get(2, std::max<long long>(tot, 1) * sizeof(double)))) { rc = 1; break; }
      if (!hip_ok(sm_alloc((void **)&d_hb, b->nPairs * sizeof(long long)), "hipMalloc")) { rc = 1; break; }
      if (!hip_ok(hipMemcpyAsync(d_hb, hb.data(), b->nPairs * sizeof(long long), hipMemcpyHostToDevice, g_stream), "H2D")) { rc = 1; break; }
      tm.start();
      rc = medium_forward_rolling(m, f->fwdSum, f->geoFS, b->d_pairs, b->pairs, b->d_in, b->d_out, d_halo, d_hb, d_ll, g_stream);
      g_last_kernel = (medium_jit_ready(f->fwdSum, MB_FORWARD, MED_MAT_NONE) || medium_jit_ready(f->fwdSum, MB_FORWARD, MED_MAT_ROLL)) ? "k_medium_jit" : "k_medium_tile<0>";
      g_last_ms += tm.stop();
      if (!rc && !hip_ok(hipStreamSynchronize(g_stream), "rolling forward kernel")) rc = 1;
    } while (0);
    if (d_hb) sm_free(d_hb);
  } else if (mode == MB_FORWARD && !b->hasEnv && use_medium(m) && env_int("MB_MEDIUM_PIPELINE", 1) &&
             (size_t)b->totalCells * 8 > budget_bytes()) {
    // the matrices of the batch do not fit the device-memory budget together:
    // ForwardMatrix semantics with only logLike() kept: continuous pipeline over recycled matrix slots
    FastState *f = fast_state(m);
    const size_t budget = budget_bytes();
    const long long want = std::min<long long>(b->totalCells, (long long)(budget / 8));
    double *pool = (double *)ws_get(0, (size_t)std::max<long long>(want, b->maxPairCells) * sizeof(double));
    if (!pool) rc = 1;
    else if ((long long)(g_ws[0].bytes / 8) < b->maxPairCells) { set_error("a single DP matrix exceeds the device memory budget"); rc = 1; }
    else {
      tm.start();
      rc = medium_forward_pipelined(m, f->fwdSum, f->geoFS, b->pairs, b->d_in, b->d_out, pool, (long long)(g_ws[0].bytes / 8), d_ll, g_stream);
      g_last_kernel = medium_jit_ready(f->fwdSum, MB_FORWARD, MED_MAT_FULL) ? "k_medium_jit" : "k_medium_tile<0>";
      g_last_ms += tm.stop();
    }
  } else {
    std::vector<Chunk> chunks;
    if (!plan_chunks(b, 1, chunks)) { sm_free(d_ll); return 1; }
    for (const Chunk &c : chunks) {
      PairDesc *d_desc = nullptr; double *pool = nullptr;
      std::vector<PairDesc> hp;
      if ((rc = upload_chunk_descs(b, c, &d_desc, hp))) break;
      if (!(pool = (double *)ws_get(0, std::max<long long>(c.cells, 1) * sizeof(double)))) { sm_free(d_desc); rc = 1; break; }
      tm.start();
      rc = fill_chunk(m, mode, d_desc, hp, b->d_in, b->d_out, pool, 0, b);
      if (!rc) rc = launch_gather_loglike(d_desc, c.p1 - c.p0, pool, m->S, 0, d_ll + c.p0, g_stream);
      g_last_ms += tm.stop();
      if (!rc && !hip_ok(hipStreamSynchronize(g_stream), "fill kernel")) rc = 1;
      if (!rc && wide_parts_failed()) rc = 1;
      if (rc) quiesce_streams();      // (a kernel launched before the failure may still read the descriptors)
      sm_free(d_desc);
      if (rc) break;
    }
  }
  if (!rc && !hip_ok(hipMemcpy(loglike, d_ll, b->nPairs * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) rc = 1;
  sm_free(d_ll);
  return rc;
}

// A one-tape sweep with k workgroups per sequence whose exchange timed out (the device is shared, a CU mask is set: the parts were
// not co-resident) has latched its programs to one workgroup per sequence: the call is run once more instead of failing.
static int with_parts_retry(const std::function<int()> &run) {
  int rc = run();
  if (rc > 0 && wide_parts_retry()) {
    (void)hipStreamSynchronize(g_stream);
    rc = run();
  }
  if (rc) wide_parts_reset();
  return rc;
}

int mb_batch_forward(mb_batch *b, int flags, double *loglike) {
  ApiGuard guard;
  if (!b || !loglike) { set_error("null argument"); return 1; }
  return with_parts_retry([&] { return run_fill_loglike(b, MB_FORWARD, flags, loglike); });
}

// ---- Viterbi ------------------------------------------------------------------------------------------------
int64_t mb_viterbi_path_bound(const mb_machine *m, int64_t inLen, int64_t outLen) {
  if (!m) return 0;
  // between two emitting steps the traceback follows silent edges to strictly lower states through at most
  // nLevF-1 levels; one more run may precede the first emission.
  return (inLen + outLen + 1) * (int64_t)m->nLevF + 1;
}

// ViterbiMatrix over a batch on the tiled / generic families.  tb: ONE traceback byte per cell instead of the fp64 matrix (tiled
// family, run-time specialised kernel; returns -1 before anything was written when that kernel is unavailable).
// tb: 0 = fp64 Viterbi matrices, 1 = the tiled family's traceback bytes, 2 = the one-tape family's traceback codes
static int viterbi_chunks(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap, int tb) {
  const bool wantPaths = pathEdges != nullptr && pathOff != nullptr;
  const int Sb = tb == 2 ? wide_tb_stride(b->m->S) : medium_tb_stride(b->m->S);
  std::vector<Chunk> chunks;
  if (!plan_chunks(b, 1, chunks, tb ? (double)Sb / b->m->S + 0.01 : 0.0)) return 1;
  int rc = 0;
  Timer tm;
  long long written = 0;
  const bool timing = opt_env("MB_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now(), tPrev = t0;
  auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[mbhip] viterbi %-28s %7.2f ms\n", what, t - tPrev); tPrev = t; } };
  for (const Chunk &c : chunks) {
    const long long np = c.p1 - c.p0;
    PairDesc *d_desc = nullptr; double *pool = nullptr, *d_ll = nullptr;
    long long *d_slot = nullptr, *d_len = nullptr; uint32_t *d_path = nullptr;
    std::vector<long long> slot(np + 1, 0), len(np, 0);
    std::vector<double> hll(np);
    std::vector<PairDesc> hp;
    do {
      if ((rc = upload_chunk_descs(b, c, &d_desc, hp, tb ? Sb : 0))) break;
      if (!(pool = (double *)ws_get(0, tb ? (size_t)std::max<long long>(c.cells / b->m->S * Sb, 16) : std::max<long long>(c.cells, 1) * sizeof(double)))) { rc = 1; break; }
      if (!hip_ok(sm_alloc((void **)&d_ll, np * sizeof(double)), "hipMalloc")) { rc = 1; break; }
      lap("chunk set-up");
      tm.start();
      if (tb == 2) {
        WideProgram *W = wide_tb_program(b->m);
        if (!W) { rc = c.p0 > 0 ? 1 : -1; if (rc > 0) set_error("one-tape traceback-code program became unavailable mid-batch"); break; }
        if ((rc = wide_fill_tb(b->m, *W, d_desc, np, b->m->nOut ? b->d_out : b->d_in, (unsigned char *)pool, d_ll, g_stream, hp.data(), device_cus()))) break;
        g_last_kernel = wide_kernel_name(*W);      // (the generated kernel or the interpreter, one workgroup or k: asked after the launch)
      } else if (tb) {
        MedEnv me;
        if (b->hasEnv) { me.d_start = b->d_envStart; me.d_end = b->d_envEnd; me.h_start = b->h_envStart.data(); me.h_end = b->h_envEnd.data(); }
        if ((rc = launch_fill_neg_inf(d_ll, np, g_stream))) break;   // (a pair whose end cell lies in a tile that does not run)
        MedGeom *tbGeo = nullptr;
        MedProgram *tbP = medium_tb_program(b->m, &tbGeo);
        rc = tbP ? medium_viterbi_tb(b->m, *tbP, *tbGeo, d_desc, hp, b->d_in, b->d_out, (unsigned char *)pool, d_ll, g_stream, me) : -1;
        if (rc < 0 && c.p0 > 0) { set_error("traceback-byte Viterbi kernel became unavailable mid-batch"); rc = 1; }
        if (rc) break;
        g_last_kernel = "k_medium_jit";
      } else {
        if ((rc = fill_chunk(b->m, MB_VITERBI, d_desc, hp, b->d_in, b->d_out, pool, 0, b))) break;
        if ((rc = launch_gather_loglike(d_desc, np, pool, b->m->S, 0, d_ll, g_stream))) break;
      }
      if (wantPaths) {
        for (long long p = 0; p < np; ++p)
          slot[p + 1] = slot[p] + mb_viterbi_path_bound(b->m, b->pairs[c.p0 + p].inLen, b->pairs[c.p0 + p].outLen);
        // cached workspaces: a hipMalloc/hipFree pair per call costs more than the traceback kernel of a small batch
        if (!(d_slot = (long long *)ws_get(3, (np + 1) * sizeof(long long)))) { rc = 1; break; }
        if (!(d_len = (long long *)ws_get(4, np * sizeof(long long)))) { rc = 1; break; }
        if (!(d_path = (uint32_t *)ws_get(5, std::max<long long>(slot[np], 1) * sizeof(uint32_t)))) { rc = 1; break; }
        if (!hip_ok(hipMemcpyAsync(d_slot, slot.data(), (np + 1) * sizeof(long long), hipMemcpyHostToDevice, g_stream), "H2D")) { rc = 1; break; }
        if (tb == 2) { if ((rc = wide_traceback_codes(b->m, *wide_tb_program(b->m), d_desc, np, (const unsigned char *)pool, d_ll, d_slot, d_path, d_len, g_stream))) break; }
        else if (tb) { if ((rc = launch_traceback_bytes(b->m, d_desc, np, b->d_in, b->d_out, (const unsigned char *)pool, Sb, d_ll, d_slot, d_path, d_len, g_stream))) break; }
        else {
          // one-tape machines beyond the LDS edge tables of the generic walkers: columns in LDS, one trip to memory per step
          WideTbPlan *T = nullptr;
          if (wide_applicable(b->m) && g_kernel_choice != 1 && b->m->nTrans > env_int("MB_ONETAPE_TRACEBACK_MIN_TRANS", 3584) && !b->hasEnv) {
            FastState *f = fast_state(b->m);
            if (!f->wTb.tried) (void)wide_traceback_build(b->m, f->wTb);
            if (f->wTb.ok) T = &f->wTb;
          }
          if (T) { if ((rc = wide_traceback(b->m, *T, d_desc, np, b->m->nOut ? b->d_out : b->d_in, pool, d_slot, d_path, d_len, g_stream))) break; }
          else if ((rc = launch_traceback(b->m, d_desc, np, b->d_in, b->d_out, pool, d_slot, d_path, d_len, g_stream))) break;
        }
      }
      lap("launches (host side)");
      g_last_ms += tm.stop();
      if (!hip_ok(hipStreamSynchronize(g_stream), "viterbi kernels")) { rc = 1; break; }
      if (wide_parts_failed()) { rc = 1; break; }
      lap("kernels");
      if (!hip_ok(hipMemcpy(hll.data(), d_ll, np * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) { rc = 1; break; }
      std::memcpy(loglike + c.p0, hll.data(), np * sizeof(double));
      if (wantPaths) {
        if (!hip_ok(hipMemcpy(len.data(), d_len, np * sizeof(long long), hipMemcpyDeviceToHost), "D2H path lengths")) { rc = 1; break; }
        std::vector<long long> off(np, -1);
        long long total = 0;
        for (long long p = 0; p < np && !rc; ++p) {
          const long long n = len[p];
          if (n == -1) continue;   // -inf end cell: no path (src/dpmatrix.defs.h:84)
          if (n < 0) { set_error(n == -2 ? "Viterbi traceback exceeded its path bound" : "Viterbi traceback reached a dead end"); rc = 1; break; }
          off[p] = total; total += n;
        }
        if (rc) break;
        if (written + total > pathCap) { set_error("pathCap too small for the Viterbi paths of this batch"); rc = 1; break; }
        // packed on the device, then ONE copy of exactly the used bytes into the caller's array (the slots are sized for
        // the worst case, (inLen+outLen+1) x levels edges per pair: 32 MB for 1024 x 1 kb x 1 kb dnapsw pairs, 15 MB used)
        long long *d_off = (long long *)ws_get(6, np * sizeof(long long));
        uint32_t *d_packed = (uint32_t *)ws_get(7, std::max<long long>(total, 1) * sizeof(uint32_t));
        if (!d_off || !d_packed) { rc = 1; break; }
        if (!hip_ok(hipMemcpyAsync(d_off, off.data(), np * sizeof(long long), hipMemcpyHostToDevice, g_stream), "H2D")) { rc = 1; break; }
        if ((rc = launch_compact_paths(d_path, d_slot, d_len, d_off, d_packed, np, g_stream))) break;
        if (total && d2h_large(pathEdges + written, d_packed, total * sizeof(uint32_t))) { rc = 1; break; }
        if (!hip_ok(hipStreamSynchronize(g_stream), "path compaction")) { rc = 1; break; }
        lap("pack + D2H paths");
        for (long long p = 0; p < np; ++p) {
          if (len[p] > 0) written += len[p];
          pathOff[c.p0 + p + 1] = written;
        }
      }
    } while (0);
    void *ptrs[] = {d_desc, d_ll};
    for (void *q : ptrs) sm_free(q);
    if (rc) break;
  }
  return rc;
}

static int batch_viterbi(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap);
int mb_batch_viterbi(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap) {
  ApiGuard guard;
  if (!b || !loglike) { set_error("null argument"); return 1; }
  return with_parts_retry([&] { return batch_viterbi(b, loglike, pathOff, pathEdges, pathCap); });
}
static int batch_viterbi(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap) {
  g_last_ms = 0.0; g_last_launches = 0;
  g_last_kernel = "";
  if (pathOff) pathOff[0] = 0;
  if (b->nPairs == 0) return 0;
  if (use_small(b->m) && small_can_run(((FastState *)b->m->fast)->smF, SM_TB, false, b->hasEnv)) return small_viterbi(b, loglike, pathOff, pathEdges, pathCap);
  // tiled family: one traceback byte per cell when the machine's exact program allows it
  MedGeom *tbGeo = nullptr;
  if (!wide_applicable(b->m) && use_medium(b->m) && env_int("MB_MEDIUM_TB", 1) && medium_tb_program(b->m, &tbGeo) &&
      traceback_bytes_lds(b->m, medium_tb_stride(b->m->S))) {
    const int rc = viterbi_chunks(b, loglike, pathOff, pathEdges, pathCap, 1);
    if (rc >= 0) return rc;
    g_last_ms = 0.0; g_last_launches = 0;
    if (pathOff) pathOff[0] = 0;
  }
  // one-tape family: one traceback code per cell when paths are wanted (a fill without paths keeps the leaner fp64 sweep).
  // MB_ONETAPE_TB=0 forbids it (the walker then re-evaluates candidates on the fp64 Viterbi matrix).  The code sweep costs 13 % over
  // the plain max sweep (a compare and a select per candidate, a second butterfly over the places), its walker half of the fp64
  // walker, and it moves an eighth of the bytes: 64 x 50 kb on the 5 063-state machine 315 vs 346 ms with 16 instead of 130 GB,
  // 256 x 4 kb 28 vs 42 ms, 64 x 2 kb 13.3 vs 14.2, the whole fn3 composite (L2-resident ring) 158 vs 171 -- and 557 GB of fp64 for
  // that machine at 64 x 50 kb could not be resident at all.  (Until the reduction was split into two butterflies the sweep cost
  // 29 % and the route was taken only when memory asked for it.)
  const bool tbWanted = env_int("MB_ONETAPE_TB", 1) != 0;
  if (pathEdges && pathOff && !b->hasEnv && tbWanted && wide_tb_program(b->m)) {
    const int rc = viterbi_chunks(b, loglike, pathOff, pathEdges, pathCap, 2);
    if (rc >= 0) return rc;
    g_last_ms = 0.0; g_last_launches = 0;
    pathOff[0] = 0;
  }
  return viterbi_chunks(b, loglike, pathOff, pathEdges, pathCap, 0);
}

// ---- Forward-Backward counts --------------------------------------------------------------------------------
// roll: the tiled family's count sweep WITHOUT a Forward matrix (medium_counts_rolling) -- one matrix per pair instead of two,
// so twice the pairs per chunk; returns -1 before anything was accumulated when that kernel is unavailable
// usage3: both matrices materialised by the plain fills, then the dependency-free usage pass of mb_usage.hip (tiled family)
static int counts_chunks(mb_batch *b, double *counts, double *loglikeSum, double *loglike, bool roll, bool usage3 = false) {
  const long long nT = b->m->nTrans;
  std::vector<Chunk> chunks;
  if (!plan_chunks(b, roll ? 1 : 2, chunks)) return 1;
  double *d_counts = nullptr, *d_ll = nullptr;
  if (!hip_ok(sm_alloc((void **)&d_counts, std::max<long long>(nT, 1) * sizeof(double)), "hipMalloc(counts)")) return 1;
  if (!hip_ok(sm_alloc((void **)&d_ll, b->nPairs * sizeof(double)), "hipMalloc(loglike)") ||
      !hip_ok(hipMemsetAsync(d_counts, 0, std::max<long long>(nT, 1) * sizeof(double), g_stream), "memset(counts)")) {
    sm_free(d_counts); sm_free(d_ll);
    return 1;
  }
  int rc = 0;
  Timer tm;
  for (const Chunk &c : chunks) {
    const long long np = c.p1 - c.p0;
    PairDesc *d_desc = nullptr; double *fwd = nullptr, *bwd = nullptr;
    std::vector<PairDesc> hp;
    do {
      if ((rc = upload_chunk_descs(b, c, &d_desc, hp))) break;
      { const size_t need = (size_t)std::max<long long>(c.cells, 1) * sizeof(double); const int sl[2] = {1, 0}; const size_t nb[2] = {need, need}; ws_plan(roll ? 1 : 2, sl, nb); }
      if (!roll && !(fwd = (double *)ws_get(0, std::max<long long>(c.cells, 1) * sizeof(double)))) { rc = 1; break; }
      if (!(bwd = (double *)ws_get(1, std::max<long long>(c.cells, 1) * sizeof(double)))) { rc = 1; break; }
      long long maxc = 0;
      for (long long p = c.p0; p < c.p1; ++p)
        maxc = std::max(maxc, (long long)(b->pairs[p].inLen + 1) * (b->pairs[p].outLen + 1) * b->m->S);
      tm.start();
      // One-tape machine (one workgroup per sequence): the Backward and the Forward fill are independent and a batch of fewer
      // sequences than CUs leaves most of the chip idle, so the two run side by side on two streams (64 sequences: 64 + 64
      // CUs); the count kernel waits for both.
      bool fwdDone = false;
      // One-tape E-step over LONG sequences: the fills carry their log-sum-exp correction term in fp64 (mb_wide.hip, wide_exp64): the
      // fp32 term's per-column error repeats in stationary states and grows linearly with the length (7.6e-5 per transition at 50 000
      // columns).  MB_ONETAPE_COUNT_FP64: 1 always, 0 never, default: sequences of >= MB_ONETAPE_COUNT_FP64_MIN_LEN (10 000) symbols.
      struct AccurateFills { bool on; explicit AccurateFills(bool o) : on(o) { if (on) wide_set_accurate(true); } ~AccurateFills() { if (on) wide_set_accurate(false); } };
      int longest = 0;
      for (const PairDesc &pd : hp) longest = std::max(longest, std::max(pd.inLen, pd.outLen));
      const int accMode = env_int("MB_ONETAPE_COUNT_FP64", -1);
      const AccurateFills accurateFills(wide_applicable(b->m) && (accMode == 1 || (accMode != 0 && longest >= env_int("MB_ONETAPE_COUNT_FP64_MIN_LEN", 10000))));
      if (!roll && !b->hasEnv && wide_applicable(b->m) && g_kernel_choice != 1 && env_int("MB_ONETAPE_CONCURRENT_FILLS", 1)) {
        WideProgram *WB = wide_program(b->m, MB_BACKWARD), *WF = wide_program(b->m, MB_FORWARD);
        hipStream_t s2 = second_stream();
        if (!WB || !WF) { rc = 1; break; }
        // (with CUs to spare each fill runs k workgroups per sequence, on a stream of its own)
        // (a machine whose ring lives in L2 and whose parts fit the LDS only when a fill has the whole chip: one fill after the other)
        bool parts = s2 && wide_parts_for(b->m, *WF, np, device_cus() / 2, hp.data()) > 1 && wide_parts_for(b->m, *WB, np, device_cus() / 2, hp.data()) > 1;
        const bool oneByOne = s2 && !parts && (WF->retGv || WB->retGv) && wide_parts_for(b->m, *WF, np, device_cus(), hp.data()) > 1 && wide_parts_for(b->m, *WB, np, device_cus(), hp.data()) > 1;
        parts = parts || oneByOne;
        const int share = oneByOne ? 1 : 2;
        const int fusedFill = parts ? -1 : wide_fill2(b->m, *WF, *WB, d_desc, d_desc, np, np, b->m->nOut ? b->d_out : b->d_in, fwd, bwd, g_stream, false);   // both sweeps in ONE launch
        if (fusedFill > 0) { rc = 1; break; }
        if (fusedFill == 0) { fwdDone = true; g_last_kernel = wide_kernel_name(*WF); }
        else if (s2) {                              // (programs of different kernel variants: two launches on two streams)
          const int *tape = b->m->nOut ? b->d_out : b->d_in;
          hipEvent_t evStart = nullptr, evDone = nullptr;
          bool ok = hipEventCreateWithFlags(&evStart, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&evDone, hipEventDisableTiming) == hipSuccess;
          ok = ok && hipEventRecord(evStart, g_stream) == hipSuccess && hipStreamWaitEvent(s2, evStart, 0) == hipSuccess;   // s2 starts after what g_stream has queued (descriptors)
          if (ok) {
            rc = wide_fill(b->m, *WF, d_desc, np, tape, fwd, nullptr, oneByOne ? g_stream : s2, false, parts ? hp.data() : nullptr, device_cus() / share);
            if (!rc) rc = wide_fill(b->m, *WB, d_desc, np, tape, bwd, nullptr, g_stream, false, parts ? hp.data() : nullptr, device_cus() / share);
            ok = hipEventRecord(evDone, s2) == hipSuccess && hipStreamWaitEvent(g_stream, evDone, 0) == hipSuccess;
          }
          if (evStart) (void)hipEventDestroy(evStart);
          if (evDone) (void)hipEventDestroy(evDone);
          if (!ok) { set_error("one-tape counts: stream synchronisation failed"); rc = 1; }
          if (rc) break;
          fwdDone = true;
          g_last_kernel = wide_kernel_name(*WF);
        }
      }
      if (!fwdDone && (rc = fill_chunk(b->m, MB_BACKWARD, d_desc, hp, b->d_in, b->d_out, bwd, 0, b))) break;
      // fused path: the Forward sweep accumulates the counts while its anti-diagonals are still in LDS
      int fused = -1;
      if (roll) {
        FastState *f = fast_state(b->m);
        MedEnv me;
        if (b->hasEnv) { me.d_start = b->d_envStart; me.d_end = b->d_envEnd; me.h_start = b->h_envStart.data(); me.h_end = b->h_envEnd.data(); }
        if ((rc = launch_fill_neg_inf(d_ll + c.p0, np, g_stream))) break;   // (a pair whose end cell lies in a tile that does not run)
        fused = medium_counts_rolling(b->m, f->fwdCnt, f->geoCnt, d_desc, hp, b->d_in, b->d_out, bwd, d_counts, d_ll + c.p0, g_stream, me);
        if (fused < 0 && c.p0 == 0) { rc = -1; break; }      // kernel unavailable: the caller runs the two-matrix path
        if (fused) { if (fused < 0) set_error("count sweep kernel became unavailable mid-batch"); rc = 1; break; }
        g_last_kernel = "k_medium_jit";
      } else if (usage3) {
        // (fused stays -1: Forward fill, log-likelihoods, then the usage pass below)
      } else if (use_medium(b->m) && fast_state(b->m)->countOk && !(b->hasEnv && wide_applicable(b->m))) {
        FastState *f = fast_state(b->m);
        MedEnv me;
        if (b->hasEnv) {
          me.d_start = b->d_envStart; me.d_end = b->d_envEnd; me.h_start = b->h_envStart.data(); me.h_end = b->h_envEnd.data();
          if ((rc = launch_fill_neg_inf(fwd, c.cells, g_stream))) break;      // tiles outside the envelopes do not run
        }
        fused = medium_counts_materialised(b->m, f->fwdCnt, f->geoCnt, d_desc, hp, b->d_in, b->d_out, fwd, bwd, d_counts, d_ll + c.p0, g_stream, me);
        if (fused > 0) { rc = 1; break; }
        if (fused == 0) g_last_kernel = "k_medium_jit";
      }
      if (fused < 0) {
        if (!fwdDone && (rc = fill_chunk(b->m, MB_FORWARD, d_desc, hp, b->d_in, b->d_out, fwd, 0, b))) break;
        if ((rc = launch_gather_loglike(d_desc, np, fwd, b->m->S, 0, d_ll + c.p0, g_stream))) break;
        // one-tape machines: a lane owns a transition and walks the columns (mb_wide.hip); two tapes: one thread per cell
        const bool oneTape = ((b->m->nIn != 0) != (b->m->nOut != 0)) && g_kernel_choice != 1 && env_int("MB_ONETAPE_COUNTS", 1);
        if (usage3) {
          if ((rc = usage_launch(b->m, fast_state(b->m)->usage, d_desc, hp, b->d_in, b->d_out, fwd, bwd, d_counts, g_stream))) break;
          g_last_kernel = "k_medium_jit (two fills) + k_medium_usage";
        } else if (oneTape) {
          FastState *f = fast_state(b->m);
          if (!f->wCnt.ok && !wide_counts_build(b->m, f->wCnt)) { rc = 1; break; }
          if ((rc = wide_counts(b->m, f->wCnt, d_desc, hp, b->m->nOut ? b->d_out : b->d_in, fwd, bwd, d_counts, g_stream))) break;
          g_last_kernel = "k_onetape_counts";
        } else if ((rc = launch_generic_counts(b->m, d_desc, np, maxc, b->d_in, b->d_out, fwd, bwd, d_counts, g_stream))) break;
      }
      g_last_ms += tm.stop();
      if (!hip_ok(hipStreamSynchronize(g_stream), "counts kernels")) { rc = 1; break; }
      if (wide_parts_failed()) { rc = 1; break; }
    } while (0);
    if (rc) quiesce_streams();      // (the sweep on the second stream may still read the descriptors freed below)
    sm_free(d_desc);
    if (rc) break;
  }
  if (rc < 0) { (void)hipStreamSynchronize(g_stream); sm_free(d_counts); sm_free(d_ll); return -1; }
  if (!rc) {
    std::vector<double> hc(nT), hll(b->nPairs);
    if (nT && !hip_ok(hipMemcpy(hc.data(), d_counts, nT * sizeof(double), hipMemcpyDeviceToHost), "D2H counts")) rc = 1;
    if (!rc && !hip_ok(hipMemcpy(hll.data(), d_ll, b->nPairs * sizeof(double), hipMemcpyDeviceToHost), "D2H loglike")) rc = 1;
    if (!rc) {
      if (g_deterministic)
        for (long long e = 0; e < nT && !rc; ++e) {      // fixed point, 2^-36
          unsigned long long u; std::memcpy(&u, &hc[e], 8);
          if (!det_to_double(u, hc[e])) { set_error("MB_DETERMINISTIC: a posterior count left the fixed-point range (6.7e7 per transition and call): split the batch or use the floating-point mode"); rc = 1; }
        }
      if (rc) { sm_free(d_counts); sm_free(d_ll); return 1; }
      for (long long e = 0; e < nT; ++e) counts[e] += hc[e];
      double s = 0;
      for (long long p = 0; p < b->nPairs; ++p) { s += hll[p]; if (loglike) loglike[p] = hll[p]; }  // loglike += forward.logLike()
      if (loglikeSum) *loglikeSum += s;
    }
  }
  sm_free(d_counts); sm_free(d_ll);
  return rc;
}

static int batch_counts(mb_batch *b, double *counts, double *loglikeSum, double *loglike);
int mb_batch_counts(mb_batch *b, double *counts, double *loglikeSum, double *loglike) {
  ApiGuard guard;
  if (!b || !counts) { set_error("null argument"); return 1; }
  return with_parts_retry([&] { return batch_counts(b, counts, loglikeSum, loglike); });      // (the counts reach the caller's array after the last chunk only: a second run adds nothing twice)
}
static int batch_counts(mb_batch *b, double *counts, double *loglikeSum, double *loglike) {
  g_last_ms = 0.0; g_last_launches = 0;
  g_last_kernel = "";
  g_deterministic = env_int("MB_DETERMINISTIC", 0) != 0;
  if (b->nPairs == 0) return 0;
  if (use_small(b->m) && small_count_fits(((FastState *)b->m->fast)->smF, b->hasEnv) && small_can_run(((FastState *)b->m->fast)->smB, SM_SUM, true, b->hasEnv))
    return small_counts(b, counts, loglikeSum, loglike);
  // Tiled family, two ways: the FUSED sweep (Backward fill, then a Forward sweep that weighs the transitions while its cells are in LDS:
  // 16 B per lattice cell, one matrix per pair) or THREE passes (both fills materialised, then the dependency-free usage pass of
  // mb_usage.hip).  Measured on the 482-state composition at 24 x 487 aa x 10 kb: fused 593 ms; three passes 946 ms -- the usage pass
  // itself 286 ms, but two matrices per pair leave 6 pairs per chunk and the two fills of such a chunk cannot fill the chip (660 ms
  // where 256 pairs through the pipeline would take 280).  So the fused sweep is the default; MB_MEDIUM_COUNT_PASSES=3 asks for the three
  // passes (a host with few long pairs and memory to spare; the parity tests).
  if (!wide_applicable(b->m) && use_medium(b->m) && !b->hasEnv) {
    const int passes = env_int("MB_MEDIUM_COUNT_PASSES", 0);
    FastState *f = fast_state(b->m);
    if (f->mediumOk && passes == 3 && (size_t)b->maxPairCells * 16 <= budget_bytes()) {
      if (!f->usage.tried) usage_build(b->m, f->usage);
      if (f->usage.ok) return counts_chunks(b, counts, loglikeSum, loglike, false, true);
    }
  }
  if (!wide_applicable(b->m) && use_medium(b->m) && fast_state(b->m)->countOk && env_int("MB_MEDIUM_COUNTS_ROLL", 1)) {
    const int rc = counts_chunks(b, counts, loglikeSum, loglike, true);
    if (rc >= 0) return rc;
    g_last_ms = 0.0; g_last_launches = 0;
  }
  return counts_chunks(b, counts, loglikeSum, loglike, false);
}

// ---- single full matrix -------------------------------------------------------------------------------------
int mb_fill_env(mb_machine *m, int mode, const int32_t *in, int64_t inLen, const int32_t *out, int64_t outLen,
                int32_t startState, const int32_t *envStart, const int32_t *envEnd, double *cellsOut) {
  ApiGuard guard;
  if (!m || !cellsOut) { set_error("null argument"); return 1; }
  if (mode < MB_FORWARD || mode > MB_BACKWARD) { set_error("mb_fill: unknown mode"); return 1; }
  if (startState < 0 || startState >= m->S) { set_error("mb_fill: start state out of range"); return 1; }
  const int64_t inOff[2] = {0, inLen}, outOff[2] = {0, outLen};
  mb_batch *b = mb_batch_create(m, 1, in, inOff, out, outOff);
  if (!b) return 1;
  if (envStart && envEnd) {
    const int64_t envOff[2] = {0, outLen + 1};
    if (mb_batch_set_envelopes(b, envOff, envStart, envEnd)) { mb_batch_destroy(b); return 1; }
  }
  if (startState == 0 && use_small(m) &&
      small_can_run(mode == MB_BACKWARD ? ((FastState *)m->fast)->smB : ((FastState *)m->fast)->smF, mode == MB_VITERBI ? SM_MAX : SM_SUM, true, b->hasEnv)) {
    const int rcs = small_fill(b, mode, cellsOut);
    mb_batch_destroy(b);
    return rcs;
  }
  const long long n = b->totalCells;
  double *pool = nullptr;
  int rc = 0;
  if (n * 8ull > budget_bytes()) { set_error("matrix exceeds the device memory budget"); rc = 1; }
  if (!rc && !(pool = (double *)ws_get(0, n * sizeof(double)))) rc = 1;
  if (!rc && env_int("MB_DEBUG_POISON", 0)) (void)hipMemsetAsync(pool, 0xFF, n * sizeof(double), g_stream);
  for (int attempt = 0; attempt < 2 && !rc; ++attempt) {      // (a second time when the exchange between the parts of a one-tape sweep timed out: see with_parts_retry)
    rc = fill_chunk(m, mode, b->d_pairs, b->pairs, b->d_in, b->d_out, pool, mode == MB_FORWARD ? startState : 0, b);
    if (!rc && !hip_ok(hipStreamSynchronize(g_stream), "fill kernel")) rc = 1;
    if (!rc && wide_parts_failed()) { if (attempt == 0 && wide_parts_retry()) continue; rc = 1; }
    break;
  }
  if (rc) wide_parts_reset();
  if (!rc && !hip_ok(hipMemcpy(cellsOut, pool, n * sizeof(double), hipMemcpyDeviceToHost), "D2H matrix")) rc = 1;
  mb_batch_destroy(b);
  return rc;
}

int mb_fill(mb_machine *m, int mode, const int32_t *in, int64_t inLen, const int32_t *out, int64_t outLen,
            int32_t startState, double *cellsOut) {
  return mb_fill_env(m, mode, in, inLen, out, outLen, startState, nullptr, nullptr, cellsOut);
}

// ---- introspection: generated source of the run-time specialised tile kernel (host only, no device needed) ------
int mb_debug_jit_source(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                        const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward, int closure,
                        int G, const char *path) {
  ApiLock lock;
  if (nStates <= 0 || nTrans < 0 || !path) { set_error("mb_debug_jit_source: bad argument"); return 1; }
  if (!medium_valid_G(G)) { set_error("mb_debug_jit_source: G must be a power of two in 1..64"); return 1; }
  mb_machine m;
  m.S = nStates; m.nIn = nInTok; m.nOut = nOutTok; m.nTrans = nTrans;
  m.src.assign(src, src + nTrans); m.dst.assign(dst, dst + nTrans);
  m.inTok.assign(inTok, inTok + nTrans); m.outTok.assign(outTok, outTok + nTrans);
  m.logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(&m, &err)) { set_error(err); return 1; }
  // mode: MB_FORWARD sum, MB_VITERBI max, 3 count, 4 max with traceback bytes; + 16: tiles without a matrix (MED_MAT_ROLL; implied by 4);
  // + 32: the PROGRAM instead of the source (see below)
  const int matKind = ((mode & 16) || (mode & 15) == MED_MODE_TB) ? MED_MAT_ROLL : MED_MAT_FULL;
  const bool dumpProgram = (mode & 32) != 0;
  const bool compactRing = (mode & 64) != 0;      // + 64: the matrix-free kernel with the COMPACT ring (as many wavefronts as its LDS allows, at most 12)
  mode &= 15;
  MedProgram P; MedGeom geo;
  if (matKind == MED_MAT_ROLL) geo.haloSteps = 0;
  if (mode == MED_MODE_COUNT) {
    if (!medium_build_count_host(&m, G, P, geo, closure)) { set_error("machine does not qualify for the fused count kernel"); return 1; }
  } else if (!medium_build_host(&m, backward != 0, closure, G, P, geo)) return 1;
  if (opt_env("MB_MEDIUM_JIT_VERBOSE")) {
    const long long c = medium_program_cost(P); int syncs = 0;
    for (const MedRoundInfo &ri : P.roundInfo) syncs += ri.sync;
    fprintf(stderr, "[mbhip] program cost %lld (rounds %zu, syncs %d, pairs %d, levels %d)\n", c, P.roundInfo.size(), syncs, P.nPairs, backward ? m.nLevB : m.nLevF);
  }
  if (dumpProgram) {
    // The program as the kernels read it -- descriptors + records, what med_slow_supercell (mb_medium.hip) interprets chunk by chunk
    // and the run-time generator unrolls --, for a device-free replay (tests/test_tiled_plan.py): 16 int32 (magic 0x4D454431, S, Spad,
    // LPG, G, nChunks, nIn, nOut, seedOff, dummyOff, records, usage slots of a flat count program, 1 = backward, closure stages,
    // 1 = counting, flags: 1 flat | 2 emitting usage fused into the fill rounds | 4 two accumulator tables), nChunks x 8 int32
    // descriptors, the records (fp64 weight, srcOff, dstOff), the usage slots (int32 table, first record, placement), one int32 per
    // record: the transition it stands for, then int32 fused slots, int32 loop-time accumulators, the fused slots (table, first
    // record, placement) and the loop-time table's transitions
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("mb_debug_jit_source: cannot open output file"); return 1; }
    // usage slots: (table, first record, placement 2 = VGPRs: summed in a register, set down in the after-the-loop table; else: the
    // loop-time table) -- the usage pass of a flat count program, then the fused emit slots of its fill rounds (MedProgram::fusedEmit)
    std::vector<int32_t> flatSlots, fusedSlots;
    for (const MedRoundInfo &ri : P.roundInfo)
      for (const MedSlotInfo &sl : ri.slots) {
        if (ri.flat) { flatSlots.push_back(sl.T); flatSlots.push_back((int32_t)sl.recBase); flatSlots.push_back(sl.place); }
        else if (ri.fused && sl.T < 3) { fusedSlots.push_back(sl.T); fusedSlots.push_back((int32_t)sl.recBase); fusedSlots.push_back(sl.place); }
      }
    const int32_t head[16] = {0x4D454431, m.S, P.Spad, P.LPG, P.G, P.nChunks, m.nIn, m.nOut, (int32_t)P.dev.seedOff, (int32_t)P.dummyOff, (int32_t)P.rec.size(),
                              (int32_t)(flatSlots.size() / 3), P.backward ? 1 : 0, closure, P.counting ? 1 : 0, (P.flatCount ? 1 : 0) | (P.fusedEmit ? 2 : 0) | (P.accAllEntries ? 4 : 0)};
    const int32_t tail[2] = {(int32_t)(fusedSlots.size() / 3), (int32_t)P.accMap.size()};
    const bool ok = fwrite(head, sizeof(head), 1, f) == 1 && fwrite(P.desc.data(), 4, (size_t)P.nChunks * MED_DESC_WORDS, f) == (size_t)P.nChunks * MED_DESC_WORDS &&
                    fwrite(P.rec.data(), sizeof(MedRec), P.rec.size(), f) == P.rec.size() &&
                    (flatSlots.empty() || fwrite(flatSlots.data(), 4, flatSlots.size(), f) == flatSlots.size()) &&
                    fwrite(P.wref.data(), 4, P.wref.size(), f) == P.wref.size() &&      // per record: >= 0 its transition, -1 padding, <= -2 a closure pair
                    fwrite(tail, sizeof(tail), 1, f) == 1 && (fusedSlots.empty() || fwrite(fusedSlots.data(), 4, fusedSlots.size(), f) == fusedSlots.size()) &&
                    (P.accMap.empty() || fwrite(P.accMap.data(), 4, P.accMap.size(), f) == P.accMap.size());      // loop-time accumulator entry -> transition
    fclose(f);
    if (!ok) { set_error("mb_debug_jit_source: short write"); return 1; }
    return 0;
  }
  if (matKind == MED_MAT_ROLL) geo.haloSteps = 0;
  if (compactRing) {
    if (matKind != MED_MAT_ROLL || P.recC.empty()) { set_error("mb_debug_jit_source: no compact ring for this kernel kind"); return 1; }
    geo.compact = true; geo.haloSteps = 0;
    for (geo.waves = 12; geo.waves > 1; --geo.waves) {
      geo.C = geo.waves * P.G;
      if (medium_jit_lds_bytes(P, geo, mode == MED_MODE_TB ? MED_MODE_TB : MB_FORWARD) <= 160 * 1024 - 512) break;
    }
  }
  if (mode == MED_MODE_TB && !medium_tb_eligible(&m, P)) { set_error("machine does not qualify for traceback bytes on the tiled family"); return 1; }
  const std::string code = medium_jit_source(&m, P, geo, mode == MED_MODE_COUNT ? MED_MODE_COUNT : (mode == MED_MODE_TB ? MED_MODE_TB : (mode == MB_VITERBI ? MB_VITERBI : MB_FORWARD)), matKind);
  FILE *f = fopen(path, "w");
  if (!f) { set_error("mb_debug_jit_source: cannot open output file"); return 1; }
  fprintf(f, "// G=%d C=%d waves=%d ldsBytes=%zu ldsRecs=%zu rounds=%zu\n", G, geo.C, geo.waves, medium_jit_lds_bytes(P, geo),
          P.ldsImageIdx.size(), P.roundInfo.size());
  fputs(code.c_str(), f);
  fclose(f);
  return 0;
}

// The retimed program of a one-tape machine (mb_wide.hip, k_wide_retimed) as the kernel reads it, written to `path`: 12 int32
// (magic 0x52455431, lanes, slots per period, ring depth NB, doubles per ring vector, largest lag, penalty row length, penalty
// entries, period, states, 1 = ring in L2, streams) followed by (NB * slots + 8) * lanes records of 16 bytes.  Host only: the
// planner can be checked without a device (tests/test_retimed_plan.py simulates the stream and compares with the oracle).
int mb_debug_wide_retimed(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                          const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward, const char *path) {
  ApiLock lock;
  if (nStates <= 0 || nTrans < 0 || !path || (nInTok != 0) == (nOutTok != 0)) { set_error("mb_debug_wide_retimed: bad argument (one-tape machines only)"); return 1; }
  mb_machine m;
  m.S = nStates; m.nIn = nInTok; m.nOut = nOutTok; m.nTrans = nTrans;
  m.src.assign(src, src + nTrans); m.dst.assign(dst, dst + nTrans);
  m.inTok.assign(inTok, inTok + nTrans); m.outTok.assign(outTok, outTok + nTrans);
  m.logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(&m, &err)) { set_error(err); return 1; }
  WideProgram P;
  std::vector<WideRec> stream;
  // mode + 16 (with MB_VITERBI, forward): the program that keeps one traceback CODE per cell; its decode tables and the incoming
  // view's edge ids follow the record streams: int32 count + tbOff, int32 count + tbEntry, int32 count + inEid
  const bool tbCodes = (mode & 16) != 0;
  mode &= 15;
  if (tbCodes && (mode != MB_VITERBI || backward)) { set_error("mb_debug_wide_retimed: traceback codes belong to the forward max program"); return 1; }
  if (!wide_ret_host(&m, backward != 0, mode == MB_VITERBI, P, stream, tbCodes)) { set_error("machine has no retimed program"); return 1; }
  if (tbCodes && !P.tbOk) { set_error("machine does not qualify for traceback codes"); return 1; }
  FILE *f = fopen(path, "wb");
  if (!f) { set_error("mb_debug_wide_retimed: cannot open output file"); return 1; }
  const int32_t head[12] = {0x52455431, P.W, P.ret.nSlots, P.ret.NB, P.ret.NVs, P.ret.kMax, P.ret.rowLen, P.ret.nPen, P.retPeriod, nStates, P.retGv ? 1 : 0, P.ret.NB};
  bool ok = fwrite(head, sizeof(head), 1, f) == 1 && fwrite(stream.data(), sizeof(WideRec), stream.size(), f) == stream.size();
  if (ok && tbCodes) {
    const int32_t n0 = (int32_t)P.h_tbOff.size(), n1 = (int32_t)P.h_tbEntry.size(), n2 = (int32_t)m.inPerm.size();
    ok = fwrite(&n0, 4, 1, f) == 1 && fwrite(P.h_tbOff.data(), 4, (size_t)n0, f) == (size_t)n0 &&
         fwrite(&n1, 4, 1, f) == 1 && fwrite(P.h_tbEntry.data(), 4, (size_t)n1, f) == (size_t)n1 &&
         fwrite(&n2, 4, 1, f) == 1 && fwrite(m.inPerm.data(), 4, (size_t)n2, f) == (size_t)n2;
  }
  fclose(f);
  if (!ok) { set_error("mb_debug_wide_retimed: short write"); return 1; }
  return 0;
}

// the k-part form of the retimed program (k workgroups per sequence, WidePartDev), planned on the host only: int32 magic 0x52455432,
// parts, exchange columns, states; then per part 16 int32 (lanes, slots, NB, NVs, kMax, rowLen, nPen, period, own states, imports,
// first export entry, first exchange column, exports, result entry, table words, byte offset of the second weights or 0), the table (machine
// state of every own state, then the exchange column of every import), the record streams as in mb_debug_wide_retimed and -- parts with
// two-transition candidates -- one double per record: the candidate's second weight
int mb_debug_wide_parts(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                        const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward, int k, int lanes, const char *path) {
  ApiLock lock;
  if (nStates <= 0 || nTrans < 0 || !path || (nInTok != 0) == (nOutTok != 0)) { set_error("mb_debug_wide_parts: bad argument (one-tape machines only)"); return 1; }
  mb_machine m;
  m.S = nStates; m.nIn = nInTok; m.nOut = nOutTok; m.nTrans = nTrans;
  m.src.assign(src, src + nTrans); m.dst.assign(dst, dst + nTrans);
  m.inTok.assign(inTok, inTok + nTrans); m.outTok.assign(outTok, outTok + nTrans);
  m.logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(&m, &err)) { set_error(err); return 1; }
  const bool tbCodes = (mode & 16) != 0;
  mode &= 15;
  std::vector<WidePartHost> parts;
  int nExpTot = 0;
  std::vector<int> tbOff; std::vector<uint32_t> tbEntry;
  if (tbCodes && (mode != MB_VITERBI || backward)) { set_error("mb_debug_wide_parts: traceback codes belong to the forward max program"); return 1; }
  int lanesChosen = lanes, ringChosen = 8;
  if (!wide_parts_host(&m, backward != 0, mode == MB_VITERBI, tbCodes, k, lanes, parts, nExpTot, &tbOff, &tbEntry, nullptr, &lanesChosen, &ringChosen)) { set_error("machine has no k-part retimed program"); return 1; }
  lanes = lanesChosen;
  FILE *f = fopen(path, "wb");
  if (!f) { set_error("mb_debug_wide_parts: cannot open output file"); return 1; }
  const int32_t head[4] = {0x52455432, (int32_t)parts.size(), nExpTot, nStates};
  bool ok = fwrite(head, sizeof(head), 1, f) == 1;
  for (const WidePartHost &H : parts) {
    const int32_t ph[16] = {lanes, H.h.ret.nSlots, H.h.ret.NB, H.h.ret.NVs, H.h.ret.kMax, H.h.ret.rowLen, H.h.ret.nPen, H.period, H.h.Sloc, H.h.nImp,
                            H.h.expBase, H.h.expIdx0, H.h.nExp, H.h.resultEntry, (int32_t)H.tab.size(), H.h.w2Offset};
    ok = ok && fwrite(ph, sizeof(ph), 1, f) == 1 && fwrite(H.tab.data(), 4, H.tab.size(), f) == H.tab.size() &&
         fwrite(H.stream.data(), sizeof(WideRec), H.stream.size(), f) == H.stream.size();
  }
  if (ok && tbCodes) {      // the joined decode tables of the parts' candidate lists, and the incoming view's edge ids
    const int32_t n0 = (int32_t)tbOff.size(), n1 = (int32_t)tbEntry.size(), n2 = (int32_t)m.inPerm.size();
    ok = fwrite(&n0, 4, 1, f) == 1 && fwrite(tbOff.data(), 4, (size_t)n0, f) == (size_t)n0 &&
         fwrite(&n1, 4, 1, f) == 1 && fwrite(tbEntry.data(), 4, (size_t)n1, f) == (size_t)n1 &&
         fwrite(&n2, 4, 1, f) == 1 && fwrite(m.inPerm.data(), 4, (size_t)n2, f) == (size_t)n2;
  }
  fclose(f);
  if (!ok) { set_error("mb_debug_wide_parts: short write"); return 1; }
  return 0;
}

// the one-tape sweep GENERATED for this machine (mb_wide_jit.cpp), planned and rendered on the host only: k >= 2: the machine cut for k
// workgroups per sequence (lanes = 0: searched), k <= 1: the one-workgroup program.  mode: MB_FORWARD / MB_VITERBI, + 16 traceback codes,
// + 64 the fp64 correction term.  Writes the HIP source to `path` and, to `path`.prog, what the source unrolls -- for the device-free replay
// of tests/test_retimed_plan.py: int32 magic 0x4A495431, parts, exchange columns, states; per part 24 int32 (lanes, NB, NVs, kMax, rowLen,
// nPen, nImp, S, expBase, nExp, expIdx0, resultEntry, slots, rounds, NPT, U, penBase, tokBase, ringBase, dummyAddr, ldsBytes, fields,
// words, 0), per slot 2 int32 (anyPen, anyW2), per round 8 int32 (firstSlot, depth, sync, uniform, gAll, anyMixed, resultLane, mask of the lane-group sizes its switch serves), per
// field 4 int32 (kind, index, cm, words), then the table [words][lanes] uint32.  compile != 0: the source is also compiled with hiprtc
// (no device needed) and the call fails when the kernel would use scratch memory.
int mb_debug_wide_jit(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                      const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward, int k, int lanes, int compile, const char *path) {
  ApiLock lock;
  if (nStates <= 0 || nTrans < 0 || !path || (nInTok != 0) == (nOutTok != 0)) { set_error("mb_debug_wide_jit: bad argument (one-tape machines only)"); return 1; }
  mb_machine m;
  m.S = nStates; m.nIn = nInTok; m.nOut = nOutTok; m.nTrans = nTrans;
  m.src.assign(src, src + nTrans); m.dst.assign(dst, dst + nTrans);
  m.inTok.assign(inTok, inTok + nTrans); m.outTok.assign(outTok, outTok + nTrans);
  m.logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(&m, &err)) { set_error(err); return 1; }
  const bool tbCodes = (mode & 16) != 0, acc = (mode & 64) != 0;
  mode &= 15;
  if (tbCodes && (mode != MB_VITERBI || backward)) { set_error("mb_debug_wide_jit: traceback codes belong to the forward max program"); return 1; }
  std::vector<WidePartHost> parts;
  std::vector<WideRec> stream;
  WideProgram P;
  std::vector<WideJitIn> ins;
  int nExpTot = 0;
  if (k >= 2) {
    int lanesChosen = lanes, mergeFlag = 0;
    if (!wide_parts_host(&m, backward != 0, mode == MB_VITERBI, tbCodes, k, lanes, parts, nExpTot, nullptr, nullptr, nullptr, &lanesChosen, &mergeFlag)) { set_error("machine has no k-part retimed program"); return 1; }
    for (const WidePartHost &H : parts) {
      WideJitIn in;
      in.ret = H.h.ret; in.W = lanesChosen; in.stream = H.stream.data();
      in.w2 = H.h.w2Offset ? (const double *)((const char *)H.stream.data() + H.h.w2Offset) : nullptr;
      in.part = true; in.S = H.h.Sloc; in.Sg = nStates; in.nImp = H.h.nImp; in.expBase = H.h.expBase; in.nExp = H.h.nExp; in.expIdx0 = H.h.expIdx0; in.resultEntry = H.h.resultEntry;
      in.gmap = H.tab.data();
      ins.push_back(in);
    }
  } else {
    if (!wide_ret_host(&m, backward != 0, mode == MB_VITERBI, P, stream, tbCodes) || P.retGv) { set_error("machine has no retimed program with its ring in LDS"); return 1; }
    WideJitIn in;
    in.ret = P.ret; in.W = P.W; in.stream = stream.data(); in.S = nStates; in.Sg = nStates; in.resultEntry = backward ? 0 : nStates - 1;
    ins.push_back(in);
  }
  WideJitFlags F; F.viterbi = mode == MB_VITERBI; F.tb = tbCodes; F.acc = acc; F.backward = backward != 0; F.inputTape = nOutTok == 0; F.nExpTot = nExpTot;
  // the plan the library would run: the register estimate's choice, planned again with fewer constants in registers while the compiled
  // kernel spills (mb_wide_jit.cpp, wide_jit_plan; MB_WIDE_JIT_ATTEMPT: the attempt to start from -- the device-free replays of the later ones)
  std::vector<WideJitDesc> descs;
  std::string code, obj;
  for (int attempt = std::max(0, std::min(env_int("MB_WIDE_JIT_ATTEMPT", 0), WIDE_JIT_ATTEMPTS - 1));; ++attempt) {
    std::string why;
    if (!wide_jit_plan(ins, acc != 0, attempt, descs, &why)) { set_error("mb_debug_wide_jit: " + why); return 1; }
    code = wide_jit_source(descs, F);
    if (!compile) break;
    std::string log;
    if (!jit_compile(code, "mb_wide_jit.hip", obj, &log, nullptr)) { set_error("mb_debug_wide_jit: hiprtc: " + log.substr(0, 2000)); return 1; }
    const long long scratch = jit_kernel_meta(obj, ".private_segment_fixed_size");
    if (scratch <= 0) break;
    if (attempt + 1 >= WIDE_JIT_ATTEMPTS) { set_error("mb_debug_wide_jit: the kernel uses " + std::to_string(scratch) + " bytes of scratch memory"); return 1; }
  }
  FILE *f = fopen(path, "w");
  if (!f) { set_error("mb_debug_wide_jit: cannot open output file"); return 1; }
  fputs(code.c_str(), f);
  fclose(f);
  f = fopen((std::string(path) + ".prog").c_str(), "wb");
  if (!f) { set_error("mb_debug_wide_jit: cannot open output file"); return 1; }
  const int32_t head[4] = {0x4A495431, (int32_t)descs.size(), nExpTot, nStates};
  bool ok = fwrite(head, sizeof(head), 1, f) == 1;
  for (const WideJitDesc &D : descs) {
    const WideJitIn &in = D.in;
    const int32_t ph[24] = {in.W, in.ret.NB, in.ret.NVs, in.ret.kMax, in.ret.rowLen, in.ret.nPen, in.nImp, in.S, in.expBase, in.nExp, in.expIdx0, in.resultEntry,
                            D.nSlots, (int32_t)D.rounds.size(), D.NPT, D.U, (int32_t)D.penBase, (int32_t)D.tokBase, (int32_t)D.ringBase, (int32_t)D.dummyAddr, (int32_t)D.ldsBytes,
                            (int32_t)D.fields.size(), D.nWords, D.level | (D.IP << 8) | (D.ring << 20)};
    ok = ok && fwrite(ph, sizeof(ph), 1, f) == 1;
    for (const WideJitSlot &sl : D.slots) { const int32_t v[2] = {sl.anyPen, sl.anyW2}; ok = ok && fwrite(v, sizeof(v), 1, f) == 1; }
    for (const WideJitRound &R : D.rounds) {
      int32_t mask = 0;      // the lane-group sizes the round's switch has a case for
      for (int g : R.gWaves) mask |= g;
      const int32_t v[8] = {R.firstSlot, R.depth, R.sync, R.uniform, R.gAll, R.anyMixed, R.resultLane, mask};
      ok = ok && fwrite(v, sizeof(v), 1, f) == 1;
    }
    for (const WideJitField &fd : D.fields) { const int32_t v[4] = {fd.kind, fd.index, fd.cm, fd.words}; ok = ok && fwrite(v, sizeof(v), 1, f) == 1; }
    std::vector<uint32_t> tab, stream;
    wide_jit_table(D, F, tab, &stream);
    ok = ok && fwrite(tab.data(), 4, tab.size(), f) == tab.size();
    if (D.level >= 1) ok = ok && fwrite(stream.data(), 4, stream.size(), f) == stream.size();      // [NB][IP][lanes] packed address words
    if (in.part) ok = ok && fwrite(in.gmap + in.S, 4, (size_t)in.nImp, f) == (size_t)in.nImp;      // the exchange columns of the imports
  }
  fclose(f);
  if (!ok) { set_error("mb_debug_wide_jit: short write"); return 1; }
  if (compile) {
    if (FILE *g = fopen((std::string(path) + ".co").c_str(), "wb")) { fwrite(obj.data(), 1, obj.size(), g); fclose(g); }
  }
  return 0;
}

// generated source of the small-machine family's sweep (mode: 0 sum, 1 max, 2 traceback bytes, 3 counts); host only
int mb_debug_small_source(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                          const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward,
                          int materialise, const char *path) {
  ApiLock lock;
  const bool dumpProgram = (mode & 32) != 0;      // the PROGRAM instead of the source (see below)
  mode &= 31;
  if (nStates <= 0 || nTrans < 0 || !path || mode < 0 || mode >= SM_NMODE) { set_error("mb_debug_small_source: bad argument"); return 1; }
  mb_machine m;
  m.S = nStates; m.nIn = nInTok; m.nOut = nOutTok; m.nTrans = nTrans;
  m.src.assign(src, src + nTrans); m.dst.assign(dst, dst + nTrans);
  m.inTok.assign(inTok, inTok + nTrans); m.outTok.assign(outTok, outTok + nTrans);
  m.logW.assign(logWeight, logWeight + nTrans);
  std::string err;
  if (!compile_machine(&m, &err)) { set_error(err); return 1; }
  SmallProgram P;
  if (!small_build_host(&m, backward != 0, P)) { set_error("machine does not qualify for the small-machine family"); return 1; }
  if (dumpProgram) {
    // what the generator unrolls, for a device-free replay (tests/test_small_plan.py): 20 int32 (magic 0x534D5031, S, nIn, nOut,
    // backward, seed state, end state, table entries, off[0..3], nTab[0..3], candidates, 3 x 0), the evaluation order [S], decOff
    // [S + 1], the candidates (T, src, dup, tab) in the reference's enumeration order, then w[] (fp64) and eid[] (int32)
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("mb_debug_small_source: cannot open output file"); return 1; }
    std::vector<int32_t> cands;
    for (int d = 0; d < P.S; ++d) for (const SmSlot &sl : P.cand[d]) { cands.push_back(sl.T); cands.push_back(sl.src); cands.push_back(sl.dup); cands.push_back(sl.tab); }
    const int32_t head[20] = {0x534D5031, P.S, P.nIn, P.nOut, P.backward ? 1 : 0, P.seedState, P.endState, (int32_t)P.nEntries, (int32_t)P.off[0], (int32_t)P.off[1],
                              (int32_t)P.off[2], (int32_t)P.off[3], P.nTab[0], P.nTab[1], P.nTab[2], P.nTab[3], (int32_t)(cands.size() / 4), 0, 0, 0};
    const bool ok = fwrite(head, sizeof(head), 1, f) == 1 && fwrite(P.order.data(), 4, P.order.size(), f) == P.order.size() &&
                    fwrite(P.decOff.data(), 4, P.decOff.size(), f) == P.decOff.size() && fwrite(cands.data(), 4, cands.size(), f) == cands.size() &&
                    fwrite(P.w.data(), 8, P.w.size(), f) == P.w.size() && fwrite(P.eid.data(), 4, P.eid.size(), f) == P.eid.size();
    fclose(f);
    if (!ok) { set_error("mb_debug_small_source: short write"); return 1; }
    return 0;
  }
  const std::string code = small_jit_source(P, mode, (materialise & 1) != 0, (materialise & 2) != 0);   // bit 1: the restricted-envelope variant
  FILE *f = fopen(path, "w");
  if (!f) { set_error("mb_debug_small_source: cannot open output file"); return 1; }
  fprintf(f, "// ldsBytes=%zu H=%d NBD=%d tables: silent %d input %d output %d match %d\n", small_jit_lds_bytes(P, mode), P.H, P.NBD,
          P.nTab[3], P.nTab[1], P.nTab[2], P.nTab[0]);
  fputs(code.c_str(), f);
  fclose(f);
  return 0;
}

// ---- host-buffer convenience wrappers -------------------------------------------------------------------------
int mb_forward_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff, const int32_t *outTok,
                     const int64_t *outOff, int flags, double *loglike) {
  mb_batch *b = mb_batch_create(m, nPairs, inTok, inOff, outTok, outOff);
  if (!b) return 1;
  const int rc = mb_batch_forward(b, flags, loglike);
  mb_batch_destroy(b);
  return rc;
}

int mb_viterbi_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff, const int32_t *outTok,
                     const int64_t *outOff, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap) {
  mb_batch *b = mb_batch_create(m, nPairs, inTok, inOff, outTok, outOff);
  if (!b) return 1;
  const int rc = mb_batch_viterbi(b, loglike, pathOff, pathEdges, pathCap);
  mb_batch_destroy(b);
  return rc;
}

int mb_counts_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff, const int32_t *outTok,
                    const int64_t *outOff, double *counts, double *loglikeSum, double *loglike) {
  mb_batch *b = mb_batch_create(m, nPairs, inTok, inOff, outTok, outOff);
  if (!b) return 1;
  const int rc = mb_batch_counts(b, counts, loglikeSum, loglike);
  mb_batch_destroy(b);
  return rc;
}

}  // extern "C"

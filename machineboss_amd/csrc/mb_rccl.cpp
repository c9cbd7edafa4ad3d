// mb_rccl.cpp -- the one exchange step of the path: MachineCounts::operator+= across processes (src/counts.cpp:66-71),
// i.e. the sum of the E-step statistics of `boss --train` / `--counts` when the pair list is sharded over GPUs.
//
// One all-reduce (sum, fp64) of nTransitions + 1 values per EM iteration over RCCL (xGMI inside a node).  The RCCL
// library is opened on first use (dlopen), so that a single-GPU caller never needs it and a Python caller that already
// has torch's copy loaded keeps using that one.  A C++ host (the reference is one) bootstraps the communicator with
// mb_comm_unique_id on rank 0, ships the 128 bytes to the other ranks by whatever it already has (MPI, a file, a socket),
// and calls mb_comm_init everywhere; a Python host uses torch.distributed instead (machineboss_amd/shard.py).
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <thread>

#include "mb_internal.h"

namespace {
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*CommDestroyFn)(void *);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*ErrStrFn)(int);
struct Rccl {
  void *lib = nullptr;
  GetUniqueIdFn getUniqueId = nullptr;
  CommInitRankFn commInitRank = nullptr;
  CommDestroyFn commDestroy = nullptr;
  AllReduceFn allReduce = nullptr;
  ErrStrFn errStr = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.lib) return true;
  // RCCL must run on the HIP runtime THIS library runs on (it is handed our stream and our device buffers).  A process may
  // hold two: PyTorch-ROCm preloads its own libamdhip64 by path, and when it is imported after this library has initialised
  // the GPU on the system runtime, its copy never sees the device -- an RCCL bound to it fails ("no ROCm-capable device").  So
  // the RCCL next to OUR runtime is tried first (dladdr on a HIP entry point tells which one that is), then the usual names.
  std::string besideHip[2];
  Dl_info info;
  if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
    const std::string path(info.dli_fname);
    const size_t slash = path.rfind('/');
    if (slash != std::string::npos) { besideHip[0] = path.substr(0, slash) + "/librccl.so.1"; besideHip[1] = path.substr(0, slash) + "/librccl.so"; }
  }
  const char *names[] = {besideHip[0].c_str(), besideHip[1].c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!*n) continue;
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) { mb::set_error(std::string("RCCL library not found: ") + dlerror()); return false; }
  g_rccl.getUniqueId = (GetUniqueIdFn)dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.commInitRank = (CommInitRankFn)dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.commDestroy = (CommDestroyFn)dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.allReduce = (AllReduceFn)dlsym(g_rccl.lib, "ncclAllReduce");
  g_rccl.errStr = (ErrStrFn)dlsym(g_rccl.lib, "ncclGetErrorString");
  if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.commDestroy || !g_rccl.allReduce) {
    mb::set_error("RCCL library lacks the expected entry points");
    g_rccl.lib = nullptr;
    return false;
  }
  return true;
}

bool rccl_ok(int rc, const char *what) {
  if (rc == 0) return true;
  mb::set_error(std::string(what) + ": " + (g_rccl.errStr ? g_rccl.errStr(rc) : "RCCL error"));
  return false;
}
}  // namespace

extern "C" {

int mb_comm_unique_id(char id[128]) {
  if (!id) { mb::set_error("null argument"); return 1; }
  if (!rccl_load()) return 1;
  UniqueId u;
  if (!rccl_ok(g_rccl.getUniqueId(&u), "ncclGetUniqueId")) return 1;
  std::memcpy(id, u.internal, 128);
  return 0;
}

// ncclCommInitRank returns when ALL nRanks ranks have called it; a rank that never arrives (it crashed before this point, it
// was handed another id, its device is gone) would leave the others waiting for ever.  The wait is therefore BOUNDED
// (MB_COMM_TIMEOUT_S, default 180 s): the bootstrap runs on a helper thread bound to this thread's device; if it has not come
// back in time this call fails with a message naming the rank, and the host is expected to print mb_last_error() and EXIT with
// a non-zero code -- never to re-exec or retry in a process that has touched the GPU (bench.py, boss.py and shard.RankGroup do
// exactly that; the helper thread is abandoned with the process).
mb_comm *mb_comm_init(const char id[128], int nRanks, int rank) {
  if (!id || nRanks < 1 || rank < 0 || rank >= nRanks) { mb::set_error("mb_comm_init: bad argument"); return nullptr; }
  if (!rccl_load()) return nullptr;
  UniqueId u;
  std::memcpy(u.internal, id, 128);
  int dev = 0;
  if (!mb::hip_ok(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
  const char *ts = mb::opt_env("MB_COMM_TIMEOUT_S");
  const double timeout = (ts && *ts) ? atof(ts) : 180.0;
  struct Result { void *comm = nullptr; int rc = 0; bool deviceOk = true; };
  auto prom = std::make_shared<std::promise<Result>>();      // shared: the helper may outlive this call
  std::future<Result> fut = prom->get_future();
  const CommInitRankFn initFn = g_rccl.commInitRank;
  std::thread([prom, initFn, u, nRanks, rank, dev]() {
    Result r;
    r.deviceOk = hipSetDevice(dev) == hipSuccess;      // the device binding is per thread
    if (r.deviceOk) r.rc = initFn(&r.comm, nRanks, u, rank);
    prom->set_value(r);
  }).detach();
  if (timeout > 0 && fut.wait_for(std::chrono::duration<double>(timeout)) != std::future_status::ready) {
    mb::set_error("mb_comm_init: rank " + std::to_string(rank) + " of " + std::to_string(nRanks) + " (device " + std::to_string(dev) + "): the RCCL communicator was not formed within " +
                  std::to_string((int)timeout) + " s -- not every rank reached ncclCommInitRank with this id (MB_COMM_TIMEOUT_S); exit, do not retry in this process");
    return nullptr;
  }
  const Result r = fut.get();
  if (!r.deviceOk) { mb::set_error("mb_comm_init: rank " + std::to_string(rank) + ": hipSetDevice(" + std::to_string(dev) + ") failed on the bootstrap thread"); return nullptr; }
  if (!rccl_ok(r.rc, ("ncclCommInitRank (rank " + std::to_string(rank) + " of " + std::to_string(nRanks) + ", device " + std::to_string(dev) + ")").c_str())) return nullptr;
  return (mb_comm *)r.comm;
}

void mb_comm_destroy(mb_comm *comm) {
  if (comm && g_rccl.commDestroy) (void)g_rccl.commDestroy((void *)comm);
}

int mb_allreduce_counts(mb_comm *comm, double *counts, size_t n, double *loglike) {
  if (!comm) return 0;   // single process: the counts are already complete
  if (!counts && n) { mb::set_error("null argument"); return 1; }
  if (!rccl_load()) return 1;
  const size_t total = n + (loglike ? 1 : 0);
  if (!total) return 0;
  double *d = nullptr;
  MB_HIP(mb::sm_alloc((void **)&d, total * sizeof(double)));   // cached by size class: no hipMalloc / hipFree per EM iteration
  int rc = 0;
  do {
    if (n && !mb::hip_ok(hipMemcpyAsync(d, counts, n * sizeof(double), hipMemcpyHostToDevice, mb::g_stream), "H2D counts")) { rc = 1; break; }
    if (loglike && !mb::hip_ok(hipMemcpyAsync(d + n, loglike, sizeof(double), hipMemcpyHostToDevice, mb::g_stream), "H2D loglike")) { rc = 1; break; }
    if (!rccl_ok(g_rccl.allReduce(d, d, total, /*ncclDouble*/ 8, /*ncclSum*/ 0, (void *)comm, mb::g_stream), "ncclAllReduce")) { rc = 1; break; }
    if (n && !mb::hip_ok(hipMemcpyAsync(counts, d, n * sizeof(double), hipMemcpyDeviceToHost, mb::g_stream), "D2H counts")) { rc = 1; break; }
    if (loglike && !mb::hip_ok(hipMemcpyAsync(loglike, d + n, sizeof(double), hipMemcpyDeviceToHost, mb::g_stream), "D2H loglike")) { rc = 1; break; }
    if (!mb::hip_ok(hipStreamSynchronize(mb::g_stream), "all-reduce of counts")) { rc = 1; break; }
  } while (0);
  mb::sm_free(d);
  return rc;
}

}  // extern "C"

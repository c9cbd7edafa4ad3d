// mb_jit.cpp -- hiprtc front end with an on-disk code-object cache (see mb_jit.h).
#include "mb_jit.h"
#include "mb_internal.h"

#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mb {

static double g_jit_ms = 0.0;
static long long g_jit_compiles = 0, g_jit_hits = 0;
double jit_compile_ms() { return g_jit_ms; }
long long jit_compiles() { return g_jit_compiles; }
long long jit_cache_hits() { return g_jit_hits; }

static const char *kOpts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-munsafe-fp-atomics"};
static const int kNOpts = 5;
// MB_JIT_EXTRA_OPTS: further space-separated hiprtc options (compiler experiments: "-mllvm -amdgpu-..."); part of the cache key
// + what the caller asks for the compiles that follow (jit_more_opts; same role in the key)
static std::string g_moreOpts;
void jit_more_opts(const char *opts) { g_moreOpts = opts ? opts : ""; }
static std::vector<std::string> extra_opts() {
  std::vector<std::string> v;
  const char *e = opt_env("MB_JIT_EXTRA_OPTS");
  const std::string all = std::string(e ? e : "") + " " + g_moreOpts;
  std::string cur;
  for (const char *p = all.c_str();; ++p) {
    if (*p == ' ' || *p == 0) { if (!cur.empty()) v.push_back(cur); cur.clear(); if (!*p) break; }
    else cur += *p;
  }
  return v;
}

static unsigned long long fnv1a(const void *p, size_t n, unsigned long long h) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t k = 0; k < n; ++k) { h ^= b[k]; h *= 1099511628211ull; }
  return h;
}

static bool mkdir_p(const std::string &dir) {
  std::string cur;
  for (size_t k = 0; k <= dir.size(); ++k) {
    if (k == dir.size() || dir[k] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0700) != 0 && errno != EEXIST) return false;
    }
    if (k < dir.size()) cur += dir[k];
  }
  return true;
}

// A cache directory is used only if it is a real directory (not a symlink) that belongs to this user and that nobody else
// can write to: a code object is executable GPU code, so a directory somebody else could pre-create (the /tmp fallback)
// or write into must never be trusted.
static bool dir_is_private(const std::string &d) {
  struct stat st;
  if (lstat(d.c_str(), &st) != 0) return false;
  return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && access(d.c_str(), W_OK) == 0;
}

// read on every compile: mb_set_option("MB_JIT_CACHE", "0") or a new MB_JIT_CACHE_DIR takes effect at once
static std::string cache_dir() {
  const char *off = opt_env("MB_JIT_CACHE");
  if (off && *off == '0') return "";
  std::string d;
  if (const char *e = opt_env("MB_JIT_CACHE_DIR")) d = e;
  else if (const char *x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/mbhip";
  else if (const char *h = getenv("HOME")) d = std::string(h) + "/.cache/mbhip";
  static std::string validated;      // the last directory that passed (mkdir + ownership check are not repeated for it)
  if (!d.empty() && d == validated) return d;
  if (d.empty() || !mkdir_p(d) || !dir_is_private(d)) {
    d = "/tmp/mbhip-cache-" + std::to_string((long long)getuid());
    if (d == validated) return d;
    if (!mkdir_p(d) || !dir_is_private(d)) return "";
  }
  validated = d;
  return d;
}

static std::string cache_path(const std::string &src) {
  const std::string dir = cache_dir();
  if (dir.empty()) return "";
  unsigned long long h = 1469598103934665603ull;
  h = fnv1a(src.data(), src.size(), h);
  for (int k = 0; k < kNOpts; ++k) h = fnv1a(kOpts[k], strlen(kOpts[k]) + 1, h);
  for (const std::string &o : extra_opts()) h = fnv1a(o.c_str(), o.size() + 1, h);
  int vmaj = 0, vmin = 0;
  (void)hiprtcVersion(&vmaj, &vmin);
  h = fnv1a(&vmaj, sizeof(vmaj), h); h = fnv1a(&vmin, sizeof(vmin), h);
  char fname[64];
  snprintf(fname, sizeof(fname), "/%016llx-%zu.co", h, src.size());
  return dir + fname;
}

void jit_evict(const std::string &src) {
  const std::string path = cache_path(src);
  if (!path.empty()) unlink(path.c_str());
}

bool jit_compile(const std::string &src, const char *name, std::string &code, std::string *log, bool *fromCache) {
  if (fromCache) *fromCache = false;
  const std::string path = cache_path(src);
  if (!path.empty()) {
    if (FILE *f = fopen(path.c_str(), "rb")) {
      fseek(f, 0, SEEK_END);
      const long n = ftell(f);
      fseek(f, 0, SEEK_SET);
      if (n > 0) {
        code.assign((size_t)n, 0);
        const bool ok = fread(&code[0], 1, (size_t)n, f) == (size_t)n;
        fclose(f);
        if (ok) { ++g_jit_hits; if (fromCache) *fromCache = true; return true; }
      } else fclose(f);
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), name, 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    if (log) *log = "hiprtcCreateProgram failed";
    static bool told = false;
    if (!told) { told = true; fprintf(stderr, "[mbhip] WARNING: hiprtc is unusable (hiprtcCreateProgram failed) -- the ahead-of-time interpreter kernels run instead, 2-4 x slower\n"); }
    return false;
  }
  const std::vector<std::string> extra = extra_opts();
  std::vector<const char *> opts(kOpts, kOpts + kNOpts);
  for (const std::string &o : extra) opts.push_back(o.c_str());
  const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string lg(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &lg[0]);
    if (log) *log = lg;
    hiprtcDestroyProgram(&prog);
    // never silently: the kernel families fall back to their ahead-of-time interpreters, which are 2-4 x slower
    static bool told = false;
    if (!told) {
      told = true;
      fprintf(stderr, "[mbhip] WARNING: run-time compilation of %s failed (hiprtc: %.200s%s) -- the ahead-of-time interpreter kernels run instead, 2-4 x slower\n", name, lg.c_str(), lg.size() > 200 ? " ..." : "");
    }
    return false;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code.assign(cs, 0);
  hiprtcGetCode(prog, &code[0]);
  hiprtcDestroyProgram(&prog);
  g_jit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ++g_jit_compiles;
  if (!path.empty()) {   // write to a private name, then rename: readers never see a partial file
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    if (FILE *f = fopen(tmp.c_str(), "wb")) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      fclose(f);
      if (!ok || rename(tmp.c_str(), path.c_str()) != 0) unlink(tmp.c_str());
    }
  }
  return true;
}

long long jit_kernel_meta(const std::string &code, const char *key) {
  const size_t klen = strlen(key);
  const size_t p = code.find(key);
  if (p == std::string::npos || p + klen >= code.size()) return -1;
  const unsigned char *q = (const unsigned char *)code.data() + p + klen;
  const size_t left = code.size() - (p + klen);
  if (q[0] <= 0x7f) return q[0];
  if (q[0] == 0xcc && left >= 2) return q[1];
  if (q[0] == 0xcd && left >= 3) return ((long long)q[1] << 8) | q[2];
  if (q[0] == 0xce && left >= 5) return ((long long)q[1] << 24) | ((long long)q[2] << 16) | ((long long)q[3] << 8) | q[4];
  return -1;
}

}  // namespace mb

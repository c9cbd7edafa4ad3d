// mb_jit.h -- run-time compilation of generated HIP source with hiprtc, shared by the kernel families that specialise a
// kernel per machine (mb_medium_jit.cpp, mb_small.cpp), with an on-disk cache of the code objects.
#pragma once
#include <string>

namespace mb {

// Compiles `src` for gfx950 and returns the code object in `code`.  The result is cached on disk under
// $MB_JIT_CACHE_DIR (default: $XDG_CACHE_HOME/mbhip, ~/.cache/mbhip, else /tmp/mbhip-cache-<uid>), keyed by a hash of the
// source text, the compile options and the hiprtc version, so that a second process (or a second machine with the same
// topology and geometry) pays no compile.  MB_JIT_CACHE=0 disables the cache (both
// are read on every compile); a directory that is a symlink, belongs to someone else or is group/world-writable is not used.  *fromCache tells which happened.
bool jit_compile(const std::string &src, const char *name, std::string &code, std::string *log, bool *fromCache);

// Further space-separated hiprtc options for the compiles (and cache lookups, evictions) that follow, until called again with
// nullptr.  One use: a kernel that ends up with scratch memory is compiled again with "-mllvm -amdgpu-spill-vgpr-to-agpr=0"
// (mb_medium_jit.cpp: round 4 met a kernel of 600+ spilled VGPRs whose AGPR spill slots came back wrong under ROCm 7.2).
void jit_more_opts(const char *opts);

// Removes the cached code object of `src` (a cached file that hipModuleLoadData rejects -- truncated, or written by another
// compiler build behind the same version number -- must not latch an error: the caller evicts it and compiles again).
void jit_evict(const std::string &src);

// cumulative wall-clock milliseconds this process spent inside hiprtc, and the number of compiles / cache hits
double jit_compile_ms();
long long jit_compiles();
long long jit_cache_hits();

// value of an unsigned-integer field of the (single) kernel's metadata note in a code object, e.g. ".vgpr_spill_count",
// ".private_segment_fixed_size" (msgpack: positive fixint, or 0xcc/0xcd/0xce + big-endian uint8/16/32); -1 if missing
long long jit_kernel_meta(const std::string &codeObject, const char *key);
}  // namespace mb

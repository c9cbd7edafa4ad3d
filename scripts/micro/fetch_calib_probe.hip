// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the count sweep's read shape (MI355X_MICROARCH.md, HBM: "calibrate on
// a known byte count in your own access pattern"): every wavefront streams its own region with 16 B per lane, four
// instructions 1 KB apart per step, lanes in REVERSED order (the Backward matrix is stored in the reversed frame).
// Kernel k_fwd reads in lane order, k_rev in reversed lane order, k_split in two runs from two different 4 KB blocks
// (lanes <= r from one block, the others from the next) -- each reads exactly `bytes` once.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const double *pool, int steps, double *sink) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double *base = pool + (long long)wave * steps * 512;
  d2 acc = {0.0, 0.0};
  for (int t = 0; t < steps; ++t) {
    int l = lane, tt = t;
    if (MODE == 1) l = 63 - lane;
    if (MODE == 2) { const int r = 40; if (lane <= r) l = r - lane; else { l = r + 64 - lane; tt = (t + 1 < steps) ? t + 1 : 0; } }
    const d2 *p = (const d2 *)(base + (long long)tt * 512) + l;
    acc += p[0]; acc += p[64]; acc += p[128]; acc += p[192];
  }
  if (acc.x == 12345.678) sink[0] = acc.y;
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  const int steps = 64, nWaves = (int)(bytes / ((size_t)steps * 4096));
  double *pool, *sink; CK(hipMalloc(&pool, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(pool, 0, bytes));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_read<0>, dim3(nWaves / 4), dim3(256), 0, 0, pool, steps, sink);
  hipLaunchKernelGGL(k_read<1>, dim3(nWaves / 4), dim3(256), 0, 0, pool, steps, sink);
  hipLaunchKernelGGL(k_read<2>, dim3(nWaves / 4), dim3(256), 0, 0, pool, steps, sink);
  CK(hipDeviceSynchronize());
  printf("each kernel read %zu bytes\n", bytes);
  return 0;
}

// Store ceiling of the small-machine family's tile-major matrix layout: every wavefront streams its own tile (TS steps of
// 64 lanes x 64 B, chunk-major: 4 store instructions of 1 KB each per step), thousands of tiles at once.
//   arg1: waves per workgroup (default 4), arg2: steps per tile (default 64), arg3: fraction of lanes active in % (100)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_tiles(double *pool, int TS, int nTiles, int activeLanes, int work) {
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tile >= nTiles) return;
  double *base = pool + (long long)tile * TS * 512;
  double x = lane * 1e-3;
  for (int t = 0; t < TS; ++t) {
    for (int k = 0; k < work; ++k) x = x * 1.0000001 + 1e-9;      // stand-in for the step's arithmetic (dependent fp64 chain)
    if (lane < activeLanes) {
      d2 *p = (d2 *)(base + (long long)t * 512) + lane;
      const d2 v = {x, (double)t};
      p[0] = v; p[64] = v; p[128] = v; p[192] = v;
    }
  }
}

int main(int argc, char **argv) {
  const int TS = argc > 2 ? atoi(argv[2]) : 64;
  const int act = argc > 3 ? atoi(argv[3]) * 64 / 100 : 64;
  const size_t bytes = (size_t)8 << 30;
  const int nTiles = (int)(bytes / ((size_t)TS * 4096));
  double *pool; CK(hipMalloc(&pool, bytes));
  CK(hipMemset(pool, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int work : {0, 25, 50, 100}) {
    for (int chunk : {4096, 16384, nTiles}) {       // tiles per launch
      float best = 1e9;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        for (int t0 = 0; t0 < nTiles; t0 += chunk) {
          const int n = (nTiles - t0 < chunk) ? nTiles - t0 : chunk;
          hipLaunchKernelGGL(k_tiles, dim3((n + 3) / 4), dim3(256), 0, 0, pool + (size_t)t0 * TS * 512, TS, n, act, work);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("TS %d active %d/64 work %3d tiles/launch %6d: %.3f ms  %.1f GB/s stored, %.1f GB/s slots\n", TS, act, work, chunk, best,
             (double)bytes * act / 64 / best / 1e6, (double)bytes / best / 1e6);
    }
  }
  return 0;
}
